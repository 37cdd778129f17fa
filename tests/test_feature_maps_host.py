"""Host logic of mass_amd.nn.update_feature_maps (no GPU): which layers a call selects, which of them may share one
library call, and that nothing falls back to the CPU.  The reference's caller is the loop
`for name in update_map: self.feature_maps[name].update(observations)` (navigation_policy.py:164-171)."""
import pytest
import torch


def layers():
    from mass_amd.nn.base_projection_layer import BaseProjectionLayer
    from mass_amd.nn.applications.occupancy_projection_layer import OccupancyProjectionLayer
    from mass_amd.nn.applications.semantic_projection_layer import SemanticProjectionLayer
    kw = dict(camera_height=12, camera_width=16, map_height=8, map_width=8, map_depth=8, grid_resolution=0.25)
    return dict(occupancy=OccupancyProjectionLayer(**kw), semantic=SemanticProjectionLayer(feature_size=3, **kw),
                rgb=BaseProjectionLayer(feature_size=3, **kw))


def test_selection_follows_the_references_update_map_argument():
    from mass_amd.nn.feature_maps import _select
    maps = layers()
    assert _select(maps, None) == list(maps.values())
    assert _select(maps, "semantic") == [maps["semantic"]]
    assert _select(maps, ["rgb", "occupancy"]) == [maps["rgb"], maps["occupancy"]]
    assert _select(list(maps.values()), None) == list(maps.values())
    with pytest.raises(KeyError):
        _select(maps, "depth")                       # an unknown map name fails like the reference's dict lookup
    with pytest.raises(TypeError):
        _select(list(maps.values()), "semantic")     # names need a dict


def test_only_layers_with_a_known_update_join_the_shared_call():
    from mass_amd.nn.feature_maps import _plain_update, _geometry_stamp
    from mass_amd.nn.applications.resnet_projection_layer import ResNetProjectionLayer
    from mass_amd.nn.base_projection_layer import BaseProjectionLayer
    maps = layers()
    assert all(_plain_update(lay) for lay in maps.values())
    res = ResNetProjectionLayer(camera_height=12, camera_width=16, map_height=8, map_width=8, map_depth=8,
                                grid_resolution=0.25)
    assert not _plain_update(res)                    # subsamples the depth image before its update: keeps its own call

    class Custom(BaseProjectionLayer):
        def update(self, observation):
            return super().update(observation)
    assert not _plain_update(Custom(camera_height=12, camera_width=16, map_height=8, map_width=8, map_depth=8,
                                    feature_size=2, grid_resolution=0.25))
    # the stamp of a layer's geometry changes when reset() rewrites the edges
    lay = maps["rgb"]
    before = _geometry_stamp(lay)
    lay.reset(origin_x=0.5)
    assert _geometry_stamp(lay) != before


def test_cpu_layers_fail_loudly_instead_of_falling_back():
    from mass_amd.nn import update_feature_maps
    maps = layers()
    obs = dict(position=torch.zeros(3), yaw=torch.tensor(0.0), elevation=torch.tensor(0.0), depth=torch.ones(12, 16, 1),
               semantic=torch.zeros(12, 16, 1, dtype=torch.uint8), features=torch.zeros(12, 16, 3))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        update_feature_maps(maps, obs)
    update_feature_maps({}, obs)                      # nothing selected: nothing to do

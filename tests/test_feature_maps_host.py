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


def test_pose_cache_is_replaced_and_read_as_one_entry():
    """BaseProjectionLayer._poses keeps the last single-frame pose for the other maps of a simulator step.  Two host
    threads that ask for different poses at the same time must each get the pose of their own key: the cache is one
    (key, pose) tuple (with key and pose in two dict entries, one thread's key could meet the other's pose)."""
    import threading
    import numpy as np
    import torch
    from mass_amd.nn.base_projection_layer import BaseProjectionLayer
    from mass_amd.utils.projection import spherical_to_cartesian, pack_poses
    lay = BaseProjectionLayer(camera_height=8, camera_width=8, map_height=4, map_width=4, map_depth=8)
    poses = [(torch.tensor([0.1 * k, -0.2, 0.3]), torch.tensor(0.05 * k + 0.7 * t), torch.tensor(-0.5)) for t in range(2)
             for k in range(16)]
    want = {}
    for pos, yaw, el in poses:
        both = spherical_to_cartesian(torch.stack([yaw, yaw]), torch.stack([el, el + np.pi / 2]))
        want[(float(pos[0]), float(yaw))] = pack_poses(pos.reshape(1, 3), both[:1], both[1:])
    errors = []

    def work(t):
        for rep in range(300):
            for pos, yaw, el in poses[16 * t:16 * t + 16]:
                got = lay._poses(pos, yaw, el)
                if not torch.equal(got, want[(float(pos[0]), float(yaw))]):
                    errors.append((t, float(pos[0]), float(yaw)))
    threads = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors[:3]


def test_whole_map_reductions_refuse_cpu_tensors():
    """amax_z / column_occupied / map_stats have no CPU path either: a map that is not on a HIP device raises, a map of
    the wrong rank or dtype is refused before anything is launched."""
    import pytest
    import torch
    from mass_amd.utils import reductions
    cpu_map = torch.zeros(4, 4, 8, 3)
    for fn in (reductions.amax_z, reductions.column_occupied, reductions.map_stats):
        with pytest.raises(RuntimeError, match="HIP device"):
            fn(cpu_map)

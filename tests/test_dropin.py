"""The reference's import lines resolve to the HIP-backed classes after install()."""
import sys


def test_reference_import_paths_resolve():
    import mass_amd.dropin
    names = mass_amd.dropin.install()
    assert "mass.nn.base_projection_layer" in names
    from mass.nn.applications.occupancy_projection_layer import OccupancyProjectionLayer
    from mass.nn.applications.semantic_projection_layer import SemanticProjectionLayer
    from mass.nn.base_projection_layer import BaseProjectionLayer
    from mass.utils.projection import (spherical_to_cartesian, project_camera_rays, transform_rays, bin_rays,
                                       update_feature_map)
    from slam_rcnn.nn.base_projection_layer import BaseProjectionLayer as B2
    import mass_amd.nn.base_projection_layer as ours
    assert BaseProjectionLayer is ours.BaseProjectionLayer is B2
    assert issubclass(SemanticProjectionLayer, BaseProjectionLayer) and issubclass(OccupancyProjectionLayer, BaseProjectionLayer)
    lay = SemanticProjectionLayer(camera_height=8, camera_width=8, map_height=4, map_width=4, map_depth=4, feature_size=3)
    # the surface agent.py / navigation_policy.py touch
    for attr in ("data", "bins_x", "bins_y", "bins_z", "rays", "origin_x", "origin_y", "map_height", "map_width",
                 "map_depth", "feature_size", "update", "reset", "map_to_world", "world_to_map", "top_down",
                 "clamp_to_world", "clamp_to_map", "visualize", "get_feature_map", "forward"):
        assert hasattr(lay, attr), attr
    assert tuple(lay.data.shape) == (4, 4, 4, 3)
    for m in [k for k in sys.modules if k.startswith(("mass.", "slam_rcnn")) or k in ("mass", "slam_rcnn")]:
        del sys.modules[m]

"""coordinate transforms / top_down / visualize of the layer against the
reference's recorded outputs (tests/golden/transforms_small.npz).  Plain torch on the host:
bit-exact on CPU."""
import numpy as np
import pytest
import torch

from conftest import load_golden


@pytest.fixture(scope="module")
def tf():
    return load_golden("transforms_small.npz")


def layer(tf):
    from mass_amd.nn.base_projection_layer import BaseProjectionLayer
    oy, ox, oz = tf["origin_yxz"]
    lay = BaseProjectionLayer(camera_height=48, camera_width=64, map_height=24, map_width=24, map_depth=12,
                              feature_size=4, origin_y=oy, origin_x=ox, origin_z=oz, grid_resolution=0.1)
    lay.data.copy_(torch.tensor(tf["data"]))
    return lay


def test_world_and_map_transforms(tf):
    lay = layer(tf)
    world = torch.tensor(tf["world"])
    assert np.array_equal(lay.clamp_to_world(world).numpy(), tf["clamp_to_world"])
    assert np.array_equal(lay.world_to_map(world).numpy(), tf["world_to_map"])
    assert np.array_equal(lay.world_to_map(world[:, :2]).numpy(), tf["world_to_map_xy"])
    mapc = torch.tensor(tf["map_coords"])
    assert np.array_equal(lay.clamp_to_map(mapc).numpy(), tf["clamp_to_map"])
    assert np.array_equal(lay.map_to_world(mapc).numpy(), tf["map_to_world"])
    with pytest.raises(RuntimeError):          # the reference's xy path of map_to_world fails the same way
        lay.map_to_world(mapc[:, :2])


def test_top_down_and_visualize(tf):
    lay = layer(tf)
    assert np.array_equal(lay.top_down(depth_slice=slice(0, 8)).numpy(), tf["top_down_0_8"])
    assert np.array_equal(lay.top_down(depth_slice=None).numpy(), tf["top_down_all"])
    assert np.array_equal(np.asarray(lay.visualize({}, depth_slice=slice(0, 8)), np.float32), tf["visualize"])
    assert lay.forward is not None and lay.get_feature_map() is lay.data


def test_world_to_map_agrees_with_update_binning(tf):
    """latent-defect note 6 of SURVEY 4: world_to_map's y flip is the one bin_rays applies."""
    from oracle import massref as orc
    lay = layer(tf)
    pts = torch.tensor(tf["world"])[:200]
    rays = pts.view(1, -1, 3)
    o = orc.bin_rays_dense(lay.bins_x, lay.bins_y, lay.bins_z, torch.zeros(3), rays, torch.ones(1, 200, 1))
    v = o["valid"][0].astype(bool)
    got = lay.world_to_map(pts).numpy()
    assert np.array_equal(got[v, 0], o["ind0"][0][v]) and np.array_equal(got[v, 1], o["ind1"][0][v])
    assert np.array_equal(got[v, 2], o["ind2"][0][v])

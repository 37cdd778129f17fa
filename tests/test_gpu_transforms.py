"""SURVEY 8(f3) on the device: the callers run world_to_map / map_to_world / top_down on the GPU
every step (navigation_policy.py:374,424; agent.py:330-331), so the recorded reference outputs
(tests/golden/transforms_small.npz) are checked with the layer and its inputs on cuda:0 too."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tf():
    return load_golden("transforms_small.npz")


def layer(tf, device):
    from mass_amd.nn.base_projection_layer import BaseProjectionLayer
    oy, ox, oz = tf["origin_yxz"]
    lay = BaseProjectionLayer(camera_height=48, camera_width=64, map_height=24, map_width=24, map_depth=12,
                              feature_size=4, origin_y=oy, origin_x=ox, origin_z=oz, grid_resolution=0.1).to(device)
    lay.data.copy_(torch.tensor(tf["data"]))
    return lay


def test_transforms_on_device(tf, device):
    lay = layer(tf, device)
    world = torch.tensor(tf["world"], device=device)
    got = lay.clamp_to_world(world)
    assert got.is_cuda and np.array_equal(got.cpu().numpy(), tf["clamp_to_world"])
    assert np.array_equal(lay.world_to_map(world).cpu().numpy(), tf["world_to_map"])
    assert np.array_equal(lay.world_to_map(world[:, :2]).cpu().numpy(), tf["world_to_map_xy"])
    # host inputs are moved over, like the reference's as_tensor(..., device=self.data.device)
    assert np.array_equal(lay.world_to_map(tf["world"]).cpu().numpy(), tf["world_to_map"])
    mapc = torch.tensor(tf["map_coords"], device=device)
    assert np.array_equal(lay.clamp_to_map(mapc).cpu().numpy(), tf["clamp_to_map"])
    assert np.array_equal(lay.map_to_world(mapc).cpu().numpy(), tf["map_to_world"])
    with pytest.raises(RuntimeError):
        lay.map_to_world(mapc[:, :2])
    # the clamp limits follow the edges through reset()
    lay.reset(origin_y=1.0, origin_x=-2.0, origin_z=0.5)
    lo, hi = lay._bounds()
    assert torch.equal(lo.cpu(), torch.stack([(b[0] + b[1]) / 2 for b in (lay.bins_x, lay.bins_y, lay.bins_z)]).cpu())
    assert float(lay.clamp_to_world(torch.tensor([[100.0, 100.0, 100.0]]))[0, 0]) == float(hi[0])


def test_top_down_and_visualize_on_device(tf, device):
    lay = layer(tf, device)
    assert np.array_equal(lay.top_down(depth_slice=slice(0, 8)).cpu().numpy(), tf["top_down_0_8"])
    assert np.array_equal(lay.top_down(depth_slice=None).cpu().numpy(), tf["top_down_all"])
    assert np.array_equal(np.asarray(lay.visualize({}, depth_slice=slice(0, 8)), np.float32), tf["visualize"])

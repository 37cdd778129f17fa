"""navigable_area / amax over z on the GPU against the reference's recorded
outputs and against the same torch ops on the CPU at larger sizes."""
import numpy as np
import pytest
import torch
import torch.nn.functional as functional

from conftest import load_golden

pytestmark = pytest.mark.gpu


def ref_navigable(data, padding, depth_slice, thr):
    nav = torch.norm(data, p=1, dim=3) > thr
    if depth_slice is not None:
        nav = nav[:, :, depth_slice]
    nav = torch.logical_not(nav.any(dim=2)).to(dtype=data.dtype)
    return 1 - functional.max_pool2d(1 - nav.unsqueeze(0), 2 * padding + 1, stride=1, padding=padding).squeeze(0)


def test_golden_navigable_and_amax(device):
    from mass_amd.utils.reductions import navigable_area, amax_z
    tf = load_golden("transforms_small.npz")
    data = torch.tensor(tf["data"]).to(device)
    for thr in (0.0, 0.5):
        got = navigable_area(data, padding=2, depth_slice=slice(2, 9), obstacle_threshold=thr)
        assert np.array_equal(got.cpu().numpy(), tf[f"navigable_thr{thr}"])
    assert np.array_equal(amax_z(data).cpu().numpy(), tf["amax_z"])


@pytest.mark.parametrize("shape", [(64, 48, 40, 1), (33, 65, 96, 54), (16, 16, 7, 300), (40, 40, 256, 3)])
def test_against_torch_cpu(device, shape):
    from mass_amd.utils.reductions import navigable_area, amax_z, column_occupied
    g = torch.Generator().manual_seed(sum(shape))
    data = torch.rand(*shape, generator=g) * (torch.rand(*shape[:3], 1, generator=g) < 0.02)
    d = data.to(device)
    assert torch.equal(amax_z(d).cpu(), data.amax(dim=2))
    for sl, thr, pad in ((None, 0.0, 3), (slice(4, 32), 0.0, 1), (slice(1, 5), 0.3 * shape[3] ** 0.5, 0)):
        assert torch.equal(navigable_area(d, pad, sl, thr).cpu(), ref_navigable(data, pad, sl, thr))
    assert not column_occupied(torch.zeros_like(d)).any()


@pytest.mark.parametrize("shape", [(24, 20, 32, 54), (7, 5, 9, 3), (16, 16, 8, 1), (5, 6, 256, 54)])
def test_map_stats_matches_the_torch_expressions(device, shape):
    """mf_map_stats: `(data != 0).any(-1).sum()` exactly, `data.abs().sum()` up to the 2^-24 each term is truncated to;
    the same bits on a second run (integer sums); columns that are no multiple of four floats; an all-zero map."""
    from mass_amd.utils.reductions import map_stats
    g = torch.Generator().manual_seed(sum(shape))
    data = (torch.rand(shape, generator=g) - 0.3) * (torch.rand(shape[:3] + (1,), generator=g) < 0.2)
    data[0, 0, 0, :] = 0.0
    data[1, 1, 1, -1] = 1e-9                       # a voxel whose only non-zero entry is tiny
    d = data.to(device)
    occ, s = map_stats(d)
    assert occ == int((data != 0).any(-1).sum())
    want = float(data.abs().sum(dtype=torch.float64))
    assert abs(s - want) <= data.numel() * 2.0 ** -24 + 1e-9 * want
    assert map_stats(d) == (occ, s)
    assert map_stats(torch.zeros(shape, device=device)) == (0, 0.0)


@pytest.mark.parametrize("shape", [(9, 7, 256, 54), (6, 5, 8, 5), (4, 4, 12, 8), (5, 3, 7, 6), (3, 3, 16, 64), (2, 3, 4, 255), (8, 8, 32, 4)])
def test_amax_z_float4_path_and_its_fallbacks(device, shape):
    """mf_amax_z reads a column as float4s when it starts on 16 bytes and its length is a multiple of four floats (each
    thread then sees four fixed channels: the period is C / gcd(C, 4) float4s), else float by float: channel counts with
    every gcd, a column length that is no multiple of four, negative values, a map view at an odd offset."""
    from mass_amd.utils.reductions import amax_z
    g = torch.Generator().manual_seed(sum(shape))
    data = torch.randn(*shape, generator=g)
    d = data.to(device)
    assert torch.equal(amax_z(d).cpu(), data.amax(dim=2))
    flat = torch.zeros(d.numel() + 1, device=device)
    flat[1:] = d.reshape(-1)
    off = flat[1:].view(shape)                       # starts 4 bytes past a 16-byte boundary
    assert torch.equal(amax_z(off).cpu(), data.amax(dim=2))

import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    """Fixtures are plain numpy archives: data only, no pickles."""
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


@pytest.fixture(scope="session")
def geom():
    return load_golden("geom_small.npz")


@pytest.fixture(scope="session")
def splat():
    return load_golden("splat_small.npz")


@pytest.fixture(scope="session")
def edge():
    return load_golden("edge_cases.npz")


@pytest.fixture(scope="session")
def matchfx():
    return load_golden("match_small.npz")


@pytest.fixture(scope="session")
def device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")


# geometry of the small fixtures (tools/gen_golden.py)
SMALL = dict(H=48, W=64, FOV=90.0, MAP=32, RES=0.1)
POSE_OF_FRAME = (1, 2, 5)      # splat_small frames use geom poses 1, 2, 5


def assert_map_close(got, want, rtol=1e-4, atol=1e-6, what="map"):
    """North-star tolerance for fp32 feature sums: 1e-4 relative (+1e-6 abs
    for values that are sums of ~1e-9 weights), and the occupancy pattern
    (which entries are non-zero) must match exactly."""
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    assert got.shape == want.shape
    occ_g, occ_w = got != 0, want != 0
    assert np.array_equal(occ_g, occ_w), f"{what}: occupancy differs in {int((occ_g != occ_w).sum())} entries"
    err = np.abs(got - want)
    bound = rtol * np.abs(want) + atol
    bad = err > bound
    assert not bad.any(), (f"{what}: {int(bad.sum())} entries out of tolerance, "
                           f"max abs err {err.max():.3e}, max rel {np.max(err / (np.abs(want) + 1e-30)):.3e}")


def assert_map_close_device(got, want, rtol=1e-4, atol=1e-6, what="map", slab=32):
    """assert_map_close for full-size maps (3.6 GB at 256^3 x 54): `got` is the device map,
    `want` the oracle's CPU tensor; compared on the device slab by slab in fp64."""
    import torch
    assert tuple(got.shape) == tuple(want.shape)
    occupied = 0
    for y0 in range(0, got.shape[0], slab):
        g = got[y0:y0 + slab].to(torch.float64)
        w = want[y0:y0 + slab].to(got.device).to(torch.float64)
        diff_occ = int(((g != 0) != (w != 0)).sum())
        assert diff_occ == 0, f"{what}: occupancy differs in {diff_occ} entries (rows {y0}..)"
        err = (g - w).abs()
        bad = err > rtol * w.abs() + atol
        assert not bool(bad.any()), (f"{what}: {int(bad.sum())} entries out of tolerance in rows {y0}.., "
                                     f"max abs err {float(err.max()):.3e}")
        occupied += int((w != 0).any(-1).sum())
    return occupied


def last_fuse_mode(layer, n_frames, workspace=None):
    """Which tile kernel took the most recent sequential multi-frame call on `layer` (its own workspace
    unless one is given): mass_amd._lib.MODE_TILES / MODE_DENSE / MODE_CELLS.  Waits for the stream."""
    from mass_amd import _lib
    from mass_amd.utils.projection import _grid_struct
    g = _grid_struct(layer.data, layer.bins_x, layer.bins_y, layer.bins_z)
    n_points = n_frames * layer.rays.shape[0] * layer.rays.shape[1]
    ws = workspace if workspace is not None else layer._workspace
    wptr, _ = ws.get(1, layer.data.device)
    mode = _lib.lib.mf_fuse_last_mode(g, n_points, n_frames, wptr, _lib.current_stream(layer.data.device))
    assert mode >= 0, _lib.lib.mf_last_error()
    return mode

"""Edge cases of the fuse pipeline: odd map sizes (tiles clipped on every axis, the scalar
final pass), more frames than one call takes, wide features, hostile depth values, and the
functional entry with labels / ones."""
import numpy as np
import pytest
import torch

from conftest import assert_map_close

pytestmark = pytest.mark.gpu


def run_pair(kw, frames, device, kind="dense", batch=True, iw=0.5):
    from mass_amd.nn.base_projection_layer import BaseProjectionLayer
    from mass_amd.nn.applications.semantic_projection_layer import SemanticProjectionLayer
    from mass_amd.nn.applications.occupancy_projection_layer import OccupancyProjectionLayer
    from oracle import massref as orc
    C = kw["feature_size"]
    cls = {"dense": BaseProjectionLayer, "label": SemanticProjectionLayer}.get(kind)
    if kind == "ones":
        k2 = {k: v for k, v in kw.items() if k != "feature_size"}
        lay = OccupancyProjectionLayer(interpolation_weight=iw, **k2).to(device)
    else:
        lay = cls(interpolation_weight=iw, **kw).to(device)
    ol = orc.RefProjectionLayer(interpolation_weight=iw, **kw)
    n = frames["depth"].shape[0]
    obs = dict(position=frames["position"], yaw=frames["yaw"], elevation=frames["elevation"], depth=frames["depth"])
    if kind == "dense":
        obs["features"] = frames["features"]
    if kind == "label":
        obs["semantic"] = frames["semantic"]
    if batch:
        lay.update_batch(obs)
    else:
        for t in range(n):
            o = {k: v[t] for k, v in obs.items()}
            if kind == "label":
                o["semantic"] = o["semantic"][..., None]
            lay.update(o)
    for t in range(n):
        f = (frames["features"][t] if kind == "dense" else
             torch.nn.functional.one_hot(frames["semantic"][t].long(), C).float() if kind == "label" else
             torch.ones_like(frames["depth"][t]))
        ol.update(dict(position=frames["position"][t], yaw=frames["yaw"][t], elevation=frames["elevation"][t],
                       depth=frames["depth"][t], features=f))
    assert_map_close(lay.data.cpu().numpy(), ol.data.numpy())
    return lay, ol


def random_frames(n, h, w, C, seed, spread=0.15, dmax=1.6):
    g = torch.Generator().manual_seed(seed)
    return dict(position=spread * torch.randn(n, 3, generator=g), yaw=6.28 * torch.rand(n, generator=g),
                elevation=-0.7 * torch.rand(n, generator=g) + 0.2, depth=0.1 + dmax * torch.rand(n, h, w, 1, generator=g),
                features=torch.rand(n, h, w, C, generator=g), semantic=torch.randint(0, C, (n, h, w), generator=g))


@pytest.mark.parametrize("kind,C", [("dense", 5), ("label", 54), ("ones", 1), ("dense", 2)])
def test_odd_map_sizes_clip_tiles_and_use_scalar_final_pass(device, kind, C):
    """30 x 21 x 13 voxels: no axis is a multiple of the tile, size2 % 8 != 0 -> no float4 path."""
    kw = dict(camera_height=24, camera_width=40, map_height=30, map_width=21, map_depth=13, feature_size=C,
              grid_resolution=0.1, origin_y=0.1, origin_x=-0.05, origin_z=0.02)
    run_pair(kw, random_frames(5, 24, 40, C, seed=C), device, kind=kind)
    run_pair(kw, random_frames(1, 24, 40, C, seed=C + 1), device, kind=kind, batch=False)


def test_more_frames_than_one_call_takes(device):
    """300 tiny frames in one update_batch: split into calls of <= 256 sequential frames."""
    kw = dict(camera_height=8, camera_width=12, map_height=16, map_width=16, map_depth=8, feature_size=4,
              grid_resolution=0.15)
    run_pair(kw, random_frames(300, 8, 12, 4, seed=7, dmax=1.0), device, kind="label")


def test_wide_features_1024(device):
    kw = dict(camera_height=12, camera_width=16, map_height=12, map_width=10, map_depth=8, feature_size=1024,
              grid_resolution=0.15)
    run_pair(kw, random_frames(3, 12, 16, 1024, seed=3, dmax=0.8), device, kind="dense")


def test_hostile_depth_values(device):
    """NaN / inf / negative / zero / beyond-range depths inside otherwise normal frames; depth == 0
    is VALID in the reference (projection.py:198) and puts the point at the camera."""
    fr = random_frames(3, 24, 32, 3, seed=11)
    d = fr["depth"]
    d[0, ::3, ::5] = float("nan"); d[0, 1::4, ::7] = float("inf"); d[1, ::2, ::9] = -1.0
    d[1, 5:9, 5:9] = 0.0; d[2, ::6, ::2] = 10.0; d[2, 1::6, ::2] = 10.000001; d[2, 2::6, 1::2] = 1e-30
    kw = dict(camera_height=24, camera_width=32, map_height=32, map_width=32, map_depth=16, feature_size=3,
              grid_resolution=0.1)
    lay, ol = run_pair(kw, fr, device, kind="dense")
    assert torch.isfinite(lay.data).all()


def test_all_points_in_one_voxel(device):
    """Every pixel at depth 0: 768 points on one voxel (the fixed-point W / S2 sums must not overflow)."""
    fr = random_frames(2, 24, 32, 3, seed=2)
    fr["depth"].zero_()
    kw = dict(camera_height=24, camera_width=32, map_height=16, map_width=16, map_depth=8, feature_size=3,
              grid_resolution=0.1)
    run_pair(kw, fr, device, kind="dense", iw=1.0)


def test_functional_update_feature_map_labels_and_ones(device):
    """mass.utils.projection.update_feature_map with class ids / ones instead of fp32 rows."""
    from mass_amd.utils.projection import update_feature_map
    from oracle import massref as orc
    g = torch.Generator().manual_seed(4)
    n, S, C = 5000, (20, 24, 16), 7
    ind = [torch.randint(0, s, (n,), generator=g) for s in S]
    rat = [torch.rand(n, generator=g) for _ in S]
    lab = torch.randint(0, C, (n,), generator=g)
    m = torch.rand(*S, C, generator=g)
    got = m.clone().to(device)
    update_feature_map(*[t.to(device) for t in ind], *[t.to(device) for t in rat], lab.to(device), got, 0.7)
    want = m.clone()
    orc.update_feature_map(*ind, *rat, torch.nn.functional.one_hot(lab, C).float(), want, interpolation_weight=0.7)
    assert_map_close(got.cpu().numpy(), want.numpy())
    m1 = torch.rand(*S, 1, generator=g)
    got = m1.clone().to(device)
    update_feature_map(*[t.to(device) for t in ind], *[t.to(device) for t in rat], None, got, 1.0)
    want = m1.clone()
    orc.update_feature_map(*ind, *rat, torch.ones(n, 1), want, interpolation_weight=1.0)
    assert_map_close(got.cpu().numpy(), want.numpy())
    update_feature_map(*[t[:0].to(device) for t in ind], *[t[:0].to(device) for t in rat], None, got, 1.0)   # empty


def test_stage_timing_diagnostics(device):
    """mf_profile_enable / mf_profile_read: per-stage HIP-event times of the most recent calls."""
    from mass_amd import _lib
    kw = dict(camera_height=24, camera_width=32, map_height=32, map_width=32, map_depth=16, feature_size=3,
              grid_resolution=0.1)
    from mass_amd.nn.base_projection_layer import BaseProjectionLayer
    lay = BaseProjectionLayer(**kw).to(device)
    fr = random_frames(4, 24, 32, 3, seed=1)
    ms = np.zeros(5, np.float32)
    assert _lib.lib.mf_profile_read(0, ms.ctypes.data) < 0          # nothing recorded yet
    _lib.check(_lib.lib.mf_profile_enable(1))
    for t in range(3):
        lay.update({k: v[t] for k, v in fr.items() if k != "semantic"})
    assert _lib.check(_lib.lib.mf_profile_read(2, ms.ctypes.data)) == 3
    assert (ms >= 0).all() and ms[4] >= ms[3] > 0 and abs(ms[:4].sum() - ms[4]) < 0.05 * ms[4] + 0.01
    _lib.check(_lib.lib.mf_profile_enable(0))
    lay.update({k: v[3] for k, v in fr.items() if k != "semantic"})
    assert _lib.lib.mf_profile_read(3, ms.ctypes.data) < 0          # not recorded while disabled


@pytest.mark.parametrize("kind", ["label", "ones"])
def test_long_sequence_folds_the_lazy_decay_mid_chunk(device, kind):
    """200 near-identical frames with interpolation_weight 1: the per-voxel decay product drops
    below 2^-40 every ~50 frames, so it is folded into the LDS deltas while earlier frames of the
    same chunk still have contributions pending (they must be rescaled with it)."""
    n, h, w, C = 200, 12, 16, 6
    g = torch.Generator().manual_seed(5)
    depth = (0.6 + 0.5 * torch.rand(1, h, w, 1, generator=g)).expand(n, h, w, 1).clone()
    depth += 1e-3 * torch.rand(n, h, w, 1, generator=g)                 # same voxels, slightly different weights
    fr = dict(position=torch.zeros(n, 3), yaw=torch.full((n,), 0.4), elevation=torch.full((n,), -0.3), depth=depth,
              semantic=torch.randint(0, C, (1, h, w), generator=g).expand(n, h, w).clone())
    kw = dict(camera_height=h, camera_width=w, map_height=16, map_width=16, map_depth=16, feature_size=C if kind == "label" else 1,
              grid_resolution=0.15)
    lay, ol = run_pair(kw, fr, device, kind=kind, iw=1.0)
    assert float(ol.data.max()) > 0.1


@pytest.mark.parametrize("kind,C", [("label", 9), ("ones", 1), ("dense", 3)])
def test_one_tile_takes_a_whole_frame(device, kind, C):
    """A 120x160 frame whose points all fall into a 3x3x3-voxel corner of the map: one tile holds
    more records than a work item of the single-pass kernel takes (it is cut into parts that merge
    through global scratch), every voxel receives thousands of points (exact integer sums), and a
    third of the points sit exactly on voxel boundaries (sub-unit corner weights)."""
    h, w, n = 120, 160, 3
    g = torch.Generator().manual_seed(21)
    depth = 0.02 + 0.13 * torch.rand(n, h, w, 1, generator=g)
    depth[:, ::3] = 0.1                                                    # many identical points
    fr = dict(position=torch.tensor([[0.05, 0.05, 0.05]]).repeat(n, 1), yaw=torch.tensor([0.3, 1.2, 2.9]),
              elevation=torch.tensor([-0.2, 0.1, -0.5]), depth=depth,
              features=torch.rand(n, h, w, C, generator=g) - (0.3 if kind == "dense" else 0.0),
              semantic=torch.randint(0, C, (n, h, w), generator=g))
    kw = dict(camera_height=h, camera_width=w, map_height=16, map_width=16, map_depth=16, feature_size=C,
              grid_resolution=0.1)
    run_pair(kw, fr, device, kind=kind, batch=False)                       # three single-frame updates
    run_pair(kw, fr, device, kind=kind, batch=True)                        # the same as one sequential batch

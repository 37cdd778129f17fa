"""The headline launch itself against the oracle: BASELINE.json configs[1] (64 distribution-A
480x640 frames -> 256^3 x 54, sequential, ONE mf_fuse_frames call), its merged-mode sibling,
configs[2] at full resolution on all three maps, and the 2-rank rehearsal of bench.py."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, assert_map_close_device, last_fuse_mode

pytestmark = pytest.mark.gpu

H, W, C, M = 480, 640, 54, 256
KW = dict(camera_height=H, camera_width=W, map_height=M, map_width=M, map_depth=M, grid_resolution=0.05)


def oracle_obs(fr, t):
    return dict(position=fr["position"][t], yaw=fr["yaw"][t], elevation=fr["elevation"][t], depth=fr["depth"][t],
                features=torch.nn.functional.one_hot(fr["semantic"][t].long(), C).float())


def test_headline_launch_64_frames_vs_oracle(device):
    """All 64 frames of the bench's rank-0 batch in one sequential launch (13 chunks / several
    rounds per tile) vs 64 oracle update() calls.  MF_TEST_HEADLINE_FRAMES shortens it."""
    from oracle import massref as orc
    from mass_amd.episodes import dist_a_frames
    from mass_amd.nn.applications.semantic_projection_layer import SemanticProjectionLayer
    n = int(os.environ.get("MF_TEST_HEADLINE_FRAMES", "64"))
    fr = dist_a_frames(n, seed0=0, height=H, width=W)
    lay = SemanticProjectionLayer(feature_size=C, **KW).to(device)
    lay.update_batch(dict(position=fr["position"], yaw=fr["yaw"], elevation=fr["elevation"],
                          depth=fr["depth"].to(device), semantic=fr["semantic"].to(device)), sequential=True)
    from mass_amd import _lib
    assert last_fuse_mode(lay, n) == _lib.MODE_CELLS, "unrelated frames are sparse: fuse_cells_kernel takes the launch"
    ref = orc.RefProjectionLayer(feature_size=C, **KW)
    for t in range(n):
        ref.update(oracle_obs(fr, t))
    occupied = assert_map_close_device(lay.data, ref.data, what=f"{n} frames sequential")
    assert occupied > 900_000 * min(n, 8)


@pytest.mark.parametrize("kind,iw,batches,fmt", [("label", 0.5, 2, None), ("label", 1.0, 1, None), ("ones", 0.5, 2, None),
                                                 ("label", 0.5, 2, "records"), ("ones", 0.5, 1, "records")])
def test_room_batches_fullsize_dense_kernel_vs_oracle(device, monkeypatch, kind, iw, batches, fmt):
    """The kernel behind the room-batch rate at ITS shape: 480x640 -> 256^3 x 54, sequential
    batches of 34 room frames through fuse_dense_kernel (two chunks of 32 + 2 frames per tile, the second
    batch blends onto the first), each against the oracle loop; also iw = 1 and the occupancy (ones) map."""
    from oracle import massref as orc
    from mass_amd import _lib
    from mass_amd.episodes import room_trajectory
    from mass_amd.nn.applications.occupancy_projection_layer import OccupancyProjectionLayer
    from mass_amd.nn.applications.semantic_projection_layer import SemanticProjectionLayer
    # fmt None: the probe's choice for a real scene, aggregated entries -> fuse_cells_kernel<AGG>; "records": fuse_dense_kernel
    if fmt is None:
        monkeypatch.delenv("MF_FORMAT", raising=False)
    else:
        monkeypatch.setenv("MF_FORMAT", fmt)
    n = 34
    tr = room_trajectory(batches * n, H, W, seed=2)
    kw = dict(KW, interpolation_weight=iw)
    if kind == "label":
        lay = SemanticProjectionLayer(feature_size=C, **kw).to(device)
        ref = orc.RefProjectionLayer(feature_size=C, **kw)
    else:
        lay = OccupancyProjectionLayer(**kw).to(device)
        ref = orc.RefProjectionLayer(feature_size=1, **kw)
    for b in range(batches):
        sl = slice(b * n, (b + 1) * n)
        batch = dict(position=tr["position"][sl], yaw=tr["yaw"][sl], elevation=tr["elevation"][sl],
                     depth=tr["depth"][sl].to(device))
        if kind == "label":
            batch["semantic"] = tr["semantic"][sl].to(device)
        lay.update_batch(batch, sequential=True)
        assert last_fuse_mode(lay, n) == (_lib.MODE_DENSE if fmt else _lib.MODE_CELLS_AGG)
        for t in range(sl.start, sl.stop):
            feats = (torch.nn.functional.one_hot(tr["semantic"][t].long(), C).float() if kind == "label"
                     else torch.ones(H, W, 1))
            ref.update(dict(position=tr["position"][t], yaw=tr["yaw"][t], elevation=tr["elevation"][t],
                            depth=tr["depth"][t], features=feats))
        occupied = assert_map_close_device(lay.data, ref.data, what=f"{kind} iw {iw} room batch {b}")
        assert occupied > 20_000


def test_pipelined_headline_batches_fullsize_vs_oracle(device):
    """The issue path bench.py times: FusePipeline submit / flush (stage on the side stream, commit
    alone on the main one) of two 24-frame distribution-A batches at 480x640 -> 256^3 x 54, against the oracle."""
    from oracle import massref as orc
    from mass_amd import _lib
    from mass_amd.episodes import dist_a_frames
    from mass_amd.nn.applications.semantic_projection_layer import SemanticProjectionLayer
    from mass_amd.utils.projection import FusePipeline
    n, nb = 24, 2
    fr = dist_a_frames(n * nb, seed0=300, height=H, width=W)
    lay = SemanticProjectionLayer(feature_size=C, **KW).to(device)
    pipe = FusePipeline(device)
    for b in range(nb):
        sl = slice(b * n, (b + 1) * n)
        poses = lay._poses(fr["position"][sl], fr["yaw"][sl], fr["elevation"][sl])
        pipe.submit(lay.bins_x, lay.bins_y, lay.bins_z, lay.rays, poses, fr["depth"][sl].to(device).reshape(n, H, W),
                    fr["semantic"][sl].to(device), lay.data, interpolation_weight=lay.interpolation_weight,
                    sequential=True)
    pipe.flush()
    torch.cuda.synchronize()
    assert [last_fuse_mode(lay, n, ws) for ws in pipe.ws] == [_lib.MODE_CELLS, _lib.MODE_CELLS]
    ref = orc.RefProjectionLayer(feature_size=C, **KW)
    for t in range(n * nb):
        ref.update(oracle_obs(fr, t))
    assert_map_close_device(lay.data, ref.data, what="pipelined distribution-A batches")


@pytest.mark.parametrize("n", [4, 8])
def test_merged_frames_fullsize_vs_oracle(device, n):
    """Merged batch semantics (all frames form one point set) at full size.  Four frames go
    through the single-pass kernel, eight (more than 2^21 points) through the all-integer tile kernel."""
    from oracle import massref as orc
    from mass_amd.episodes import dist_a_frames
    from mass_amd.nn.applications.semantic_projection_layer import SemanticProjectionLayer
    fr = dist_a_frames(n, seed0=100, height=H, width=W)
    lay = SemanticProjectionLayer(feature_size=C, **KW).to(device)
    lay.data.fill_(0.0625)
    lay.update_batch(dict(position=fr["position"], yaw=fr["yaw"], elevation=fr["elevation"],
                          depth=fr["depth"].to(device), semantic=fr["semantic"].to(device)), sequential=False)
    ref = orc.RefProjectionLayer(feature_size=C, **KW)
    ref.data.fill_(0.0625)
    pts = [[] for _ in range(7)]
    for t in range(n):
        rays = orc.transform_rays(ref.rays, orc.spherical_to_cartesian(fr["yaw"][t], fr["elevation"][t]),
                                  orc.spherical_to_cartesian(fr["yaw"][t], fr["elevation"][t] + np.pi / 2))
        out = orc.bin_rays(ref.bins_x, ref.bins_y, ref.bins_z, fr["position"][t], rays, fr["depth"][t],
                           torch.nn.functional.one_hot(fr["semantic"][t].long(), C).float())
        for k in range(7):
            pts[k].append(out[k])
    ix, iy, iz, rx, ry, rz, feats = (torch.cat(p) for p in pts)
    orc.update_feature_map(iy, ix, iz, ry, rx, rz, feats, ref.data, interpolation_weight=ref.interpolation_weight)
    assert_map_close_device(lay.data, ref.data, what=f"{n} frames merged")


def test_config3_fullsize_three_maps_vs_oracle(device):
    """configs[2] at its real size: a short stretch of the room trajectory at 480x640 -> 256^3,
    occupancy (ones), semantic (54 labels) and RGB (dense C = 3) maps, per-frame update()."""
    from oracle import massref as orc
    from mass_amd.episodes import room_trajectory
    from mass_amd.nn.base_projection_layer import BaseProjectionLayer
    from mass_amd.nn.applications.occupancy_projection_layer import OccupancyProjectionLayer
    from mass_amd.nn.applications.semantic_projection_layer import SemanticProjectionLayer
    n = 6
    tr = room_trajectory(n, H, W, seed=1)
    occ = OccupancyProjectionLayer(**KW).to(device)
    sem = SemanticProjectionLayer(feature_size=C, **KW).to(device)
    rgb = BaseProjectionLayer(feature_size=3, **KW).to(device)
    o_occ = orc.RefProjectionLayer(feature_size=1, **KW)
    o_sem = orc.RefProjectionLayer(feature_size=C, **KW)
    o_rgb = orc.RefProjectionLayer(feature_size=3, **KW)
    for t in range(n):
        base = dict(position=tr["position"][t], yaw=tr["yaw"][t], elevation=tr["elevation"][t], depth=tr["depth"][t])
        occ.update(base)
        sem.update(dict(base, semantic=tr["semantic"][t][..., None]))
        rgb.update(dict(base, features=tr["rgb"][t]))
        o_occ.update(dict(base, features=torch.ones_like(tr["depth"][t])))
        o_sem.update(dict(base, features=torch.nn.functional.one_hot(tr["semantic"][t].long(), C).float()))
        o_rgb.update(dict(base, features=tr["rgb"][t]))
    assert assert_map_close_device(occ.data, o_occ.data, what="occupancy") > 10_000
    assert_map_close_device(sem.data, o_sem.data, what="semantic")
    assert_map_close_device(rgb.data, o_rgb.data, what="rgb")
    # the same stretch as ONE sequential launch per map (distribution B of configs[1])
    sem2 = SemanticProjectionLayer(feature_size=C, **KW).to(device)
    sem2.update_batch(dict(position=tr["position"], yaw=tr["yaw"], elevation=tr["elevation"],
                           depth=tr["depth"].to(device), semantic=tr["semantic"].to(device)), sequential=True)
    assert_map_close_device(sem2.data, o_sem.data, what="semantic, one launch")


def run_bench(extra, env_extra=None):
    env = dict(os.environ, PYTHONPATH=ROOT, **(env_extra or {}))
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--batch", "8",
                          "--no-cpu-baseline", "--no-extras"] + extra, env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_two_ranks_share_the_gpu_and_allreduce(device):
    """`python bench.py --gpus 2` as the driver invokes it: the parent starts two rank processes
    (torch.distributed.run) before touching the GPU; with MF_BENCH_BACKEND=gloo both ranks use
    this box's one GPU.  Every rank fuses the same frames here, so the all-reduced counters must
    be exactly twice the single-process ones."""
    # (--rank-seed-stride 0: every rank, and every one of a rank's rotating batches, is the batch of seed 0)
    one = run_bench(["--gpus", "1", "--rank-seed-stride", "0"])
    two = run_bench(["--gpus", "2", "--rank-seed-stride", "0"], {"MF_BENCH_BACKEND": "gloo"})
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    for k in ("frames", "valid_points", "touched_voxels", "union_voxels"):
        assert two["metrics_allreduce"][k] == 2 * one["metrics_allreduce"][k], k
    np.testing.assert_allclose(two["metrics_allreduce"]["map_abs_sum"], 2 * one["metrics_allreduce"]["map_abs_sum"],
                               rtol=1e-4)
    assert 0 < one["roofline"]["frac"] <= 1 and 0 < one["roofline_step"]["frac"] <= 1
    assert two["value"] > 0 and two["scaling"] == "weak"

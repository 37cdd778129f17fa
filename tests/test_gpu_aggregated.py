"""Aggregated entries (bucket_agg_kernel -> fuse_cells_kernel<AGG>): the corners of a block's points are summed per
(tile, frame, voxel, class) in LDS before they are written - collision compaction for real scenes.

Every case is compared with the oracle loop of layer.update() calls (base_projection_layer.py:282-343 ->
projection.py:233-351); mf_fuse_last_mode proves that the aggregated path ran.  Covered: a room trajectory (tens of
corners per entry), unrelated frames (hardly any collisions: most corners lose their slot or are alone in it), ones
features, iw = 1, a map that is not zero, invalid class ids, tiles whose cells need several windows, more than 64
frames per call, stage / commit pairs, and bit-identical repetition."""
import pytest
import torch

from conftest import assert_map_close, last_fuse_mode
from test_gpu_cells import layers, run_both, sparse_frames

pytestmark = pytest.mark.gpu


def room_frames(n, H, W, C, seed):
    from mass_amd.episodes import room_trajectory
    tr = room_trajectory(n, H, W, seed=seed, num_classes=C)
    return {k: tr[k] for k in ("position", "yaw", "elevation", "depth", "semantic")}


@pytest.mark.parametrize("scene", ["room", "unrelated"])
@pytest.mark.parametrize("kind,C,iw", [("label", 54, 0.5), ("label", 5, 1.0), ("ones", 1, 0.5), ("ones", 1, 1.0)])
def test_aggregated_entries_match_the_oracle(device, monkeypatch, scene, kind, C, iw):
    """Two batches of 16 frames onto a 64^3 map, the second one onto what the first left."""
    from mass_amd import _lib
    monkeypatch.setenv("MF_FORMAT", "aggregated")
    H, W, M, n = 60, 80, 64, 16
    lay, ref = layers(device, kind, C, H, W, M, 0.1, iw=iw)
    fr = room_frames(2 * n, H, W, C, seed=3) if scene == "room" else sparse_frames(2 * n, H, W, C, seed=7)
    for sl in (slice(0, n), slice(n, 2 * n)):
        run_both(lay, ref, fr, sl, kind, C)
        assert last_fuse_mode(lay, n) == _lib.MODE_CELLS_AGG
    assert_map_close(lay.data.cpu().numpy(), ref.data.numpy(), what=f"{scene} {kind} C={C} iw={iw} aggregated")


def test_probe_sends_a_room_to_the_aggregated_path(device, monkeypatch):
    from mass_amd import _lib
    monkeypatch.delenv("MF_FORMAT", raising=False)
    H, W, M, C, n = 60, 80, 64, 9, 12
    lay, ref = layers(device, "label", C, H, W, M, 0.1)
    fr = room_frames(n, H, W, C, seed=2)
    run_both(lay, ref, fr, slice(0, n), "label", C)
    assert last_fuse_mode(lay, n) == _lib.MODE_CELLS_AGG
    assert_map_close(lay.data.cpu().numpy(), ref.data.numpy(), what="room, the probe's choice")


def test_invalid_class_ids_nonzero_map_and_repetition(device, monkeypatch):
    """Class ids >= C count as all-zero feature rows (they still add to W and S2); the map starts out non-zero; a
    second run on a fresh copy of the same map gives the same bits (all sums are integers)."""
    from mass_amd import _lib
    monkeypatch.setenv("MF_FORMAT", "aggregated")
    H, W, M, C, n = 60, 80, 64, 7, 10
    fr = room_frames(n, H, W, C + 3, seed=5)                     # ids up to C + 2
    g = torch.Generator().manual_seed(9)
    init = torch.rand(M, M, M, C, generator=g) * (torch.rand(M, M, M, 1, generator=g) < 0.3)
    outs = []
    for rep in range(2):
        lay, ref = layers(device, "label", C, H, W, M, 0.1)
        lay.data.copy_(init)
        ref.data.copy_(init)
        if rep == 0:
            run_both(lay, ref, fr, slice(0, n), "label", C)
            assert_map_close(lay.data.cpu().numpy(), ref.data.numpy(), what="invalid ids onto a non-zero map, aggregated")
        else:
            batch = {k: fr[k][:n] for k in ("position", "yaw", "elevation", "depth", "semantic")}
            lay.update_batch(batch, sequential=True, validate=False)
        assert last_fuse_mode(lay, n) == _lib.MODE_CELLS_AGG
        outs.append(lay.data.clone())
    assert torch.equal(outs[0], outs[1]), "aggregated entries: two runs differ"


def test_many_frames_and_windows(device, monkeypatch):
    """150 frames in one call (64-frame windows of the masks) of a camera that keeps looking at the same wall: the tiles
    of that wall need more cells than fit and take their frames in several windows."""
    from mass_amd import _lib
    monkeypatch.setenv("MF_FORMAT", "aggregated")
    H, W, M, C, n = 48, 64, 64, 5, 150
    lay, ref = layers(device, "label", C, H, W, M, 0.1)
    fr = room_frames(n, H, W, C, seed=11)
    run_both(lay, ref, fr, slice(0, n), "label", C)
    assert last_fuse_mode(lay, n) == _lib.MODE_CELLS_AGG
    assert_map_close(lay.data.cpu().numpy(), ref.data.numpy(), what="150 room frames, aggregated")


def test_stage_commit_pairs(device, monkeypatch):
    """The pipelined issue path (stage on a side stream, commit alone) with aggregated entries."""
    from mass_amd import _lib
    from mass_amd.utils.projection import FusePipeline
    monkeypatch.setenv("MF_FORMAT", "aggregated")
    H, W, M, C, n, per = 60, 80, 64, 9, 24, 8
    lay, ref = layers(device, "label", C, H, W, M, 0.1)
    fr = room_frames(n, H, W, C, seed=4)
    pipe = FusePipeline(device)
    depth, sem = fr["depth"].to(device).reshape(n, H, W), fr["semantic"].to(device)
    for a in range(0, n, per):
        sl = slice(a, a + per)
        poses = lay._poses(fr["position"][sl], fr["yaw"][sl], fr["elevation"][sl])
        pipe.submit(lay.bins_x, lay.bins_y, lay.bins_z, lay.rays, poses, depth[sl], sem[sl], lay.data,
                    interpolation_weight=lay.interpolation_weight, sequential=True)
    pipe.flush()
    assert [last_fuse_mode(lay, per, ws) for ws in pipe.ws] == [_lib.MODE_CELLS_AGG, _lib.MODE_CELLS_AGG]
    for t in range(n):
        feats = torch.nn.functional.one_hot(fr["semantic"][t].long(), C).float()
        ref.update(dict(position=fr["position"][t], yaw=fr["yaw"][t], elevation=fr["elevation"][t],
                        depth=fr["depth"][t], features=feats))
    assert_map_close(lay.data.cpu().numpy(), ref.data.numpy(), what="stage / commit pairs, aggregated")


def test_out_of_range_class_id_calls_the_batch_off(device, monkeypatch):
    """A class id outside [0, C) anywhere in a batch of room frames (the reference's one_hot raises before anything is
    written, semantic_projection_layer.py:203-209): bucket_agg_kernel<false> sees it like count_kernel does - the call raises
    (validate=True) or reports at check_labels (validate="defer"), the map is left as it was, and the layer keeps working."""
    from mass_amd import _lib
    monkeypatch.delenv("MF_FORMAT", raising=False)
    H, W, M, C, n = 60, 80, 64, 9, 8
    lay, ref = layers(device, "label", C, H, W, M, 0.1)
    fr = room_frames(2 * n, H, W, C, seed=6)
    run_both(lay, ref, fr, slice(0, n), "label", C)
    assert last_fuse_mode(lay, n) == _lib.MODE_CELLS_AGG
    before = lay.data.clone()
    bad = {k: fr[k][n:2 * n].clone() for k in ("position", "yaw", "elevation", "depth", "semantic")}
    bad["semantic"][3, H // 2, W // 3] = C + 2
    with pytest.raises(RuntimeError, match="Class values"):
        lay.update_batch(bad, sequential=True)
    assert torch.equal(lay.data, before)
    lay.update_batch(bad, sequential=True, validate="defer")
    with pytest.raises(RuntimeError, match="Class values"):
        lay.check_labels()
    assert torch.equal(lay.data, before)
    run_both(lay, ref, fr, slice(n, 2 * n), "label", C)           # the good frames of the same slice
    assert_map_close(lay.data.cpu().numpy(), ref.data.numpy(), what="after the batches that were called off")


def test_two_host_threads_batch_calls(device, monkeypatch):
    """Two host threads, each fusing batches into its own layer on its own stream - one a room (aggregated entries), one
    unrelated frames (contributions): the probe's read-back buffer is per thread, the verdicts kept for commits are per
    workspace."""
    import threading
    from mass_amd import _lib
    monkeypatch.delenv("MF_FORMAT", raising=False)
    H, W, M, C, n = 60, 80, 64, 9, 12
    cases = [("room", room_frames(n, H, W, C, seed=8), _lib.MODE_CELLS_AGG), ("unrelated", sparse_frames(n, H, W, C, seed=9), _lib.MODE_CELLS)]
    pairs = [layers(device, "label", C, H, W, M, 0.1) for _ in cases]
    errors, modes = [], [None, None]

    def work(i):
        try:
            lay, fr = pairs[i][0], cases[i][1]
            with torch.cuda.stream(torch.cuda.Stream(device)):
                for rep in range(3):
                    lay.reset()
                    for a in range(0, n, 4):
                        lay.update_batch({k: fr[k][a:a + 4] for k in ("position", "yaw", "elevation", "depth", "semantic")},
                                         sequential=True, validate=False)
                torch.cuda.current_stream().synchronize()
                modes[i] = last_fuse_mode(lay, 4)
        except Exception as exc:                         # noqa: BLE001 (reported below)
            errors.append(exc)

    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for i, (name, fr, want) in enumerate(cases):
        assert modes[i] == want
        ref = pairs[i][1]
        for t in range(n):
            feats = torch.nn.functional.one_hot(fr["semantic"][t].long(), C).float()
            ref.update(dict(position=fr["position"][t], yaw=fr["yaw"][t], elevation=fr["elevation"][t], depth=fr["depth"][t],
                            features=feats))
        assert_map_close(pairs[i][0].data.cpu().numpy(), ref.data.numpy(), what=f"{name}, thread {i}")


@pytest.mark.parametrize("C,my,mx", [(24, 64, 64), (60, 64, 64), (54, 30, 34), (3, 22, 26)])
def test_other_channel_counts_and_maps_that_are_no_multiple_of_the_tile(device, monkeypatch, C, my, mx):
    """Every float4-slot instantiation of fuse_cells_kernel<AGG> (C <= 8, 32, 56, 64) and maps whose height / width are no
    multiples of four (border tiles: rows outside the map are neither read nor written), onto a map that is not zero."""
    from oracle import massref as orc
    from mass_amd import _lib
    from mass_amd.nn.applications.semantic_projection_layer import SemanticProjectionLayer
    monkeypatch.setenv("MF_FORMAT", "aggregated")
    H, W, n = 60, 80, 10
    kw = dict(camera_height=H, camera_width=W, map_height=my, map_width=mx, map_depth=32, grid_resolution=0.1, interpolation_weight=0.5)
    lay = SemanticProjectionLayer(feature_size=C, **kw).train().to(device)
    ref = orc.RefProjectionLayer(feature_size=C, **kw)
    g = torch.Generator().manual_seed(200 + C + my)
    init = torch.rand(my, mx, 32, C, generator=g) * (torch.rand(my, mx, 32, 1, generator=g) < 0.3)
    lay.data.copy_(init)
    ref.data.copy_(init)
    fr = sparse_frames(n, H, W, C, seed=C + mx, dmax=2.5, spread=0.2) if C == 3 else room_frames(n, H, W, C, seed=C)
    run_both(lay, ref, fr, slice(0, n), "label", C)
    assert last_fuse_mode(lay, n) == _lib.MODE_CELLS_AGG
    assert_map_close(lay.data.cpu().numpy(), ref.data.numpy(), what=f"aggregated, C={C} map {my}x{mx}x32")


PROCESS_CASE = r"""
import sys, torch
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
from conftest import assert_map_close, last_fuse_mode
from test_gpu_cells import layers, sparse_frames, run_both
from test_gpu_aggregated import room_frames
from mass_amd import _lib
from mass_amd.utils.projection import FusePipeline
dev = torch.device("cuda:0")
H, W, M, C, n = 60, 80, 64, 9, 12
for scene, want in (("room", %(room)d), ("unrelated", _lib.MODE_CELLS)):
    lay, ref = layers(dev, "label", C, H, W, M, 0.1)
    fr = room_frames(n, H, W, C, seed=2) if scene == "room" else sparse_frames(n, H, W, C, seed=41)
    run_both(lay, ref, fr, slice(0, n), "label", C)
    assert last_fuse_mode(lay, n) == want, (scene, last_fuse_mode(lay, n), want)
    assert_map_close(lay.data.cpu().numpy(), ref.data.numpy(), what=scene)
    # stage / commit pairs: the commit takes the verdict of its staging call (or, without the read-back, launches every variant)
    lay2, _ = layers(dev, "label", C, H, W, M, 0.1)
    pipe = FusePipeline(dev)
    depth, sem = fr["depth"].to(dev).reshape(n, H, W), fr["semantic"].to(dev)
    for a in range(0, n, 4):
        sl = slice(a, a + 4)
        pipe.submit(lay2.bins_x, lay2.bins_y, lay2.bins_z, lay2.rays, lay2._poses(fr["position"][sl], fr["yaw"][sl], fr["elevation"][sl]),
                    depth[sl], sem[sl], lay2.data, interpolation_weight=lay2.interpolation_weight, sequential=True)
    pipe.flush()
    assert [last_fuse_mode(lay2, 4, ws) for ws in pipe.ws] == [want, want]
    assert_map_close(lay2.data.cpu().numpy(), ref.data.numpy(), what=scene + ", stage / commit pairs")
print("PROBE_OK")
"""


@pytest.mark.parametrize("env,room_mode", [({"MF_PROBE_SYNC": "0"}, 2), ({"MF_AGG": "0"}, 2), ({}, 4)])
def test_probe_read_back_switches(env, room_mode):
    """MF_PROBE_SYNC=0: the host does not wait for the probe - every variant of contributions / records is launched and
    the probe picks on the device (aggregated entries are not offered: a room goes to fuse_dense_kernel); MF_AGG=0: read
    back, records for real scenes; default: read back, aggregated entries.  Fresh processes (read once per process); plain
    calls and stage / commit pairs, rooms and unrelated frames, each against the oracle."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    e = dict(os.environ, PYTHONPATH=ROOT, **env)
    e.pop("MF_FORMAT", None)
    out = subprocess.run([sys.executable, "-c", PROCESS_CASE % dict(root=ROOT, room=room_mode)], env=e, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0 and "PROBE_OK" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])

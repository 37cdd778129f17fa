"""Real-scene batches (SURVEY 8(d) distribution B): the all-integer tile kernel on 4x4x8 tiles.

Sequential frames of class ids / ones are bucketed on 4x4x8 tiles and tile_list_kernel picks the tile
kernel from the call's own density (include/massfuse.h, mf_fuse_frames): a room trajectory runs in
fuse_dense_kernel from the first batch on.  Batches are checked against the oracle loop of layer.update()
calls (base_projection_layer.py:282-343); the oracle is the C restatement pinned by the reference's fixtures."""
import numpy as np
import pytest
import torch

from conftest import assert_map_close, assert_map_close_device, last_fuse_mode

pytestmark = pytest.mark.gpu
H, W, MAP, C = 120, 160, 96, 7


@pytest.fixture(autouse=True)
def records_format(monkeypatch):
    """This file is about fuse_dense_kernel: real scenes keep 16-byte records (by default the probe sends them to the
    aggregated entries of bucket_agg_kernel + fuse_cells_kernel: tests/test_gpu_aggregated.py)."""
    monkeypatch.setenv("MF_FORMAT", "records")


def _layers(device, kind):
    from oracle import massref as orc
    from mass_amd.nn.applications.occupancy_projection_layer import OccupancyProjectionLayer
    from mass_amd.nn.applications.semantic_projection_layer import SemanticProjectionLayer
    kw = dict(camera_height=H, camera_width=W, map_height=MAP, map_width=MAP, map_depth=MAP, grid_resolution=0.08,
              interpolation_weight=0.5)
    if kind == "ones":
        return OccupancyProjectionLayer(**kw).train().to(device), orc.RefProjectionLayer(feature_size=1, **kw)
    return (SemanticProjectionLayer(feature_size=C, **kw).train().to(device),
            orc.RefProjectionLayer(feature_size=C, **kw))


@pytest.mark.parametrize("kind", ["label", "ones"])
def test_room_batches_take_the_dense_kernel_and_match_the_oracle(device, kind):
    from mass_amd import _lib
    from mass_amd.episodes import room_trajectory
    n = 12
    tr = room_trajectory(2 * n, H, W, seed=3, num_classes=C)
    lay, ref = _layers(device, kind)
    for half in range(2):
        sl = slice(half * n, (half + 1) * n)
        batch = dict(position=tr["position"][sl], yaw=tr["yaw"][sl], elevation=tr["elevation"][sl],
                     depth=tr["depth"][sl])
        if kind == "label":
            batch["semantic"] = tr["semantic"][sl]
        lay.update_batch(batch, sequential=True)
        torch.cuda.synchronize()                         # the density words of this call have reached the host
        for t in range(sl.start, sl.stop):
            feats = (torch.nn.functional.one_hot(tr["semantic"][t].long(), C).float() if kind == "label"
                     else torch.ones(H, W, 1))
            ref.update(dict(position=tr["position"][t], yaw=tr["yaw"][t], elevation=tr["elevation"][t],
                            depth=tr["depth"][t], features=feats))
        assert_map_close(lay.data.cpu().numpy(), ref.data.numpy(), what=f"{kind} batch {half}")
        assert last_fuse_mode(lay, n) == _lib.MODE_DENSE, "a room batch is dense: fuse_dense_kernel takes it"


def test_dense_kernel_is_run_to_run_identical(device):
    """Integer sums only (W, S2 and the deltas are 64-bit fixed point): the same batch twice gives the same bits."""
    from mass_amd import _lib
    from mass_amd.episodes import room_trajectory
    tr = room_trajectory(16, H, W, seed=5, num_classes=C)
    lay, _ref = _layers(device, "label")

    def run(sl):
        lay.update_batch(dict(position=tr["position"][sl], yaw=tr["yaw"][sl], elevation=tr["elevation"][sl],
                              depth=tr["depth"][sl], semantic=tr["semantic"][sl]), sequential=True)
        torch.cuda.synchronize()

    run(slice(0, 8))
    assert last_fuse_mode(lay, 8) == _lib.MODE_DENSE
    outs = []
    for _ in range(2):
        lay.reset()
        run(slice(8, 16))
        outs.append(lay.data.clone())
    assert bool((outs[0] != 0).any())
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("iw", [0.5, 1.0])
def test_dense_kernel_three_chunks_vs_oracle(device, iw):
    """70 frames in one call = chunks of 32 + 32 + 6 frames in fuse_dense_kernel: a later chunk multiplies the
    integer deltas of the earlier ones by its own product of a (and with iw = 1 that product is often 0)."""
    from mass_amd import _lib
    from mass_amd.episodes import room_trajectory
    from oracle import massref as orc
    from mass_amd.nn.applications.semantic_projection_layer import SemanticProjectionLayer
    h, w, m, c, n = 120, 160, 48, 5, 70
    kw = dict(camera_height=h, camera_width=w, map_height=m, map_width=m, map_depth=m, grid_resolution=0.15,
              interpolation_weight=iw)
    tr = room_trajectory(2 * n, h, w, seed=7, num_classes=c)
    lay = SemanticProjectionLayer(feature_size=c, **kw).train().to(device)
    ref = orc.RefProjectionLayer(feature_size=c, **kw)
    g = torch.Generator().manual_seed(3)
    init = torch.rand(m, m, m, c, generator=g)
    lay.data.copy_(init); ref.data.copy_(init)
    for half in range(2):
        sl = slice(half * n, (half + 1) * n)
        lay.update_batch(dict(position=tr["position"][sl], yaw=tr["yaw"][sl], elevation=tr["elevation"][sl],
                              depth=tr["depth"][sl], semantic=tr["semantic"][sl]), sequential=True)
        torch.cuda.synchronize()
        for t in range(sl.start, sl.stop):
            ref.update(dict(position=tr["position"][t], yaw=tr["yaw"][t], elevation=tr["elevation"][t],
                            depth=tr["depth"][t],
                            features=torch.nn.functional.one_hot(tr["semantic"][t].long(), c).float()))
        assert_map_close(lay.data.cpu().numpy(), ref.data.numpy(), what=f"iw {iw} batch {half}")
    assert last_fuse_mode(lay, n) == _lib.MODE_DENSE


def test_pipelined_room_batches(device):
    """FusePipeline (stage on a side stream, commit in order, two workspaces) over a room trajectory, checked
    against the oracle loop; every batch runs in fuse_dense_kernel."""
    from mass_amd import _lib
    from mass_amd.episodes import room_trajectory
    from mass_amd.utils.projection import FusePipeline
    nb, per = 6, 8
    tr = room_trajectory(nb * per, H, W, seed=11, num_classes=C)
    lay, ref = _layers(device, "label")
    pipe = FusePipeline(device)
    for b in range(nb):
        sl = slice(b * per, (b + 1) * per)
        poses = lay._poses(tr["position"][sl], tr["yaw"][sl], tr["elevation"][sl])
        pipe.submit(lay.bins_x, lay.bins_y, lay.bins_z, lay.rays, poses, tr["depth"][sl].to(device).reshape(per, H, W),
                    tr["semantic"][sl].to(device), lay.data, interpolation_weight=lay.interpolation_weight, sequential=True)
    pipe.flush()
    torch.cuda.synchronize()
    assert [last_fuse_mode(lay, per, ws) for ws in pipe.ws] == [_lib.MODE_DENSE, _lib.MODE_DENSE]
    for t in range(nb * per):
        ref.update(dict(position=tr["position"][t], yaw=tr["yaw"][t], elevation=tr["elevation"][t], depth=tr["depth"][t],
                        features=torch.nn.functional.one_hot(tr["semantic"][t].long(), C).float()))
    assert_map_close(lay.data.cpu().numpy(), ref.data.numpy(), what="pipelined room batches")

"""fuse_cells_kernel: sequential frames of class ids / ones whose points are sparse (SURVEY 8(d)
distribution A) on 4x4x8 tiles with compact (voxel, frame) cells.

Every case is compared with the oracle loop of layer.update() calls (base_projection_layer.py:282-343 ->
projection.py:233-351); mf_fuse_last_mode proves which tile kernel ran.  The cases cover one window per
tile, tiles whose cells do not fit (several windows of frames, deltas rescaled by the later windows' prod a),
more than 64 frames per call (64-frame windows), iw = 1 (a = 0 exactly on points at voxel centres), a map that
is not zero, invalid class ids counted as zero rows, and bit-identical repetition."""
import numpy as np
import pytest
import torch

from conftest import assert_map_close, last_fuse_mode

pytestmark = pytest.mark.gpu


def layers(device, kind, C, H, W, M, res, iw=0.5, depth_cells=None):
    from oracle import massref as orc
    from mass_amd.nn.applications.occupancy_projection_layer import OccupancyProjectionLayer
    from mass_amd.nn.applications.semantic_projection_layer import SemanticProjectionLayer
    kw = dict(camera_height=H, camera_width=W, map_height=M, map_width=M, map_depth=depth_cells or M,
              grid_resolution=res, interpolation_weight=iw)
    if kind == "ones":
        return OccupancyProjectionLayer(**kw).train().to(device), orc.RefProjectionLayer(feature_size=1, **kw)
    return (SemanticProjectionLayer(feature_size=C, **kw).train().to(device),
            orc.RefProjectionLayer(feature_size=C, **kw))


def sparse_frames(n, H, W, C, seed, dmin=0.4, dmax=3.0, spread=0.25):
    """Unrelated frames: random depth per pixel, random poses around the origin, uniform labels."""
    g = torch.Generator().manual_seed(seed)
    return dict(position=spread * torch.randn(n, 3, generator=g), yaw=6.2831 * torch.rand(n, generator=g),
                elevation=-0.6 * torch.rand(n, generator=g),
                depth=dmin + (dmax - dmin) * torch.rand(n, H, W, 1, generator=g),
                semantic=torch.randint(0, C, (n, H, W), generator=g).to(torch.uint8))


def run_both(lay, ref, fr, sl, kind, C):
    batch = dict(position=fr["position"][sl], yaw=fr["yaw"][sl], elevation=fr["elevation"][sl], depth=fr["depth"][sl])
    if kind == "label":
        batch["semantic"] = fr["semantic"][sl]
    lay.update_batch(batch, sequential=True, **({"validate": False} if kind == "label" else {}))
    H, W = fr["depth"].shape[1:3]
    for t in range(sl.start, sl.stop):
        if kind == "label":
            lab = fr["semantic"][t].long()
            feats = torch.nn.functional.one_hot(lab.clamp(max=C), C + 1).float()[..., :C]   # ids >= C: zero rows
        else:
            feats = torch.ones(H, W, 1)
        ref.update(dict(position=fr["position"][t], yaw=fr["yaw"][t], elevation=fr["elevation"][t],
                        depth=fr["depth"][t], features=feats))


@pytest.mark.parametrize("kind,C,iw", [("label", 54, 0.5), ("label", 5, 1.0), ("ones", 1, 0.5), ("ones", 1, 1.0)])
def test_sparse_batches_take_the_cells_kernel_and_match_the_oracle(device, kind, C, iw):
    """Two batches of 24 unrelated frames onto a 64^3 map (the second one blends onto what the first left):
    a tile sees a handful of records per frame, all frames of a tile fit one window."""
    from mass_amd import _lib
    H, W, M, n = 60, 80, 64, 24
    lay, ref = layers(device, kind, C, H, W, M, 0.1, iw)
    fr = sparse_frames(2 * n, H, W, C, seed=11, dmax=3.5)
    for half in range(2):
        run_both(lay, ref, fr, slice(half * n, (half + 1) * n), kind, C)
        assert last_fuse_mode(lay, n) == _lib.MODE_CELLS
        assert_map_close(lay.data.cpu().numpy(), ref.data.numpy(), what=f"{kind} C={C} iw={iw} batch {half}")
    assert int((ref.data != 0).any(-1).sum()) > 20_000


@pytest.mark.parametrize("kind,C,iw", [("label", 7, 0.5), ("label", 54, 1.0), ("ones", 1, 0.5)])
def test_tiles_whose_cells_do_not_fit_take_their_frames_in_windows(device, kind, C, iw, monkeypatch):
    """64 frames from almost the same place onto a 32^3 map: every frame touches most voxels of the tiles in
    front of the cameras (~100 cells per tile and frame, far more than fit).  The library's probe would give such
    a call - tens of points of a pixel patch in one tile - to fuse_dense_kernel; MF_FORMAT=contributions (read per
    call) keeps it with fuse_cells_kernel, whose windows this is about.  The later windows rescale the integer
    deltas of the earlier ones; the map starts from random values."""
    from mass_amd import _lib
    monkeypatch.setenv("MF_FORMAT", "contributions")
    H, W, M, n = 48, 64, 32, 64
    lay, ref = layers(device, kind, C, H, W, M, 0.1, iw)
    g = torch.Generator().manual_seed(2)
    init = torch.rand(M, M, M, C, generator=g) * (torch.rand(M, M, M, 1, generator=g) < 0.5)
    lay.data.copy_(init)
    ref.data.copy_(init)
    fr = sparse_frames(n, H, W, C, seed=5, dmin=0.3, dmax=1.6, spread=0.05)
    run_both(lay, ref, fr, slice(0, n), kind, C)
    assert last_fuse_mode(lay, n) == _lib.MODE_CELLS
    assert_map_close(lay.data.cpu().numpy(), ref.data.numpy(), what=f"{kind} C={C} iw={iw}")


def test_more_than_64_frames_per_call_and_invalid_ids(device):
    """150 frames in one call (windows of 64 + 64 + 22 frames per tile); a few pixels carry the id C (taken as
    an all-zero feature row when the check is off: they decay what they land on, validate=False)."""
    from mass_amd import _lib
    H, W, M, C, n = 30, 40, 48, 9, 150
    lay, ref = layers(device, "label", C, H, W, M, 0.12)
    fr = sparse_frames(n, H, W, C, seed=21, dmax=2.8)
    fr["semantic"][::7, ::5, ::3] = C
    run_both(lay, ref, fr, slice(0, n), "label", C)
    assert last_fuse_mode(lay, n) == _lib.MODE_CELLS
    assert_map_close(lay.data.cpu().numpy(), ref.data.numpy(), what="150 frames")


def test_points_on_voxel_centres_and_faces(device):
    """Depths chosen so that ratios of exactly 0.5 (corner weights of exactly 0 + 1e-9, and with iw = 1 a decay
    factor of exactly 0) and points on voxel faces occur: tiny weights must survive as non-zero map entries."""
    from mass_amd import _lib
    H, W, M, C, n = 24, 32, 32, 4, 6
    lay, ref = layers(device, "label", C, H, W, M, 0.125, iw=1.0)
    g = torch.Generator().manual_seed(8)
    depth = (torch.randint(2, 14, (n, H, W, 1), generator=g).float() * 0.0625)       # multiples of half a voxel
    fr = dict(position=torch.zeros(n, 3), yaw=torch.zeros(n), elevation=torch.zeros(n), depth=depth,
              semantic=torch.randint(0, C, (n, H, W), generator=g).to(torch.uint8))
    fr["position"][:, 0] = 0.0625 * torch.arange(n)
    run_both(lay, ref, fr, slice(0, n), "label", C)
    assert last_fuse_mode(lay, n) in (_lib.MODE_CELLS, _lib.MODE_DENSE, _lib.MODE_CELLS_AGG)
    assert_map_close(lay.data.cpu().numpy(), ref.data.numpy(), what="centres and faces")


def test_cells_kernel_is_run_to_run_identical(device):
    from mass_amd import _lib
    H, W, M, C, n = 60, 80, 64, 54, 16
    lay, _ = layers(device, "label", C, H, W, M, 0.1)
    fr = sparse_frames(n, H, W, C, seed=3)
    outs = []
    for _ in range(2):
        lay.reset()
        lay.data.fill_(0.25)
        lay.update_batch(dict(position=fr["position"], yaw=fr["yaw"], elevation=fr["elevation"], depth=fr["depth"],
                              semantic=fr["semantic"]), sequential=True)
        assert last_fuse_mode(lay, n) == _lib.MODE_CELLS
        outs.append(lay.data.clone())
    assert torch.equal(outs[0], outs[1])


def test_blend_weight_above_one_goes_to_the_float_tile_kernel(device):
    """The integer kernels need 0 <= iw <= 1 (every term bounded by 1); the reference accepts any weight:
    such a call is taken by fuse_tiles_kernel on the same tiles (TileParams.mode_force)."""
    H, W, M, C, n = 48, 64, 32, 5, 5
    lay, ref = layers(device, "label", C, H, W, M, 0.1, iw=1.25)
    fr = sparse_frames(n, H, W, C, seed=13, dmax=1.5)
    run_both(lay, ref, fr, slice(0, n), "label", C)
    assert_map_close(lay.data.cpu().numpy(), ref.data.numpy(), rtol=2e-4, what="iw 1.25")


@pytest.mark.parametrize("scene", ["unrelated", "room"])
@pytest.mark.parametrize("fmt", ["contributions", "records", "aggregated", None])
def test_either_entry_format_gives_the_oracle_map(device, monkeypatch, scene, fmt):
    """The probe's choice of the tile-local entry format (contributions -> fuse_cells_kernel, aggregated entries ->
    fuse_cells_kernel<AGG>; 16-byte records -> fuse_dense_kernel when forced or with MF_AGG=0) only decides the speed:
    forced any way, or left to the probe (None), a batch of unrelated frames and a room trajectory both come out within
    tolerance of the oracle; left alone, the probe sends the unrelated frames to contributions and the room to
    aggregated entries."""
    from mass_amd import _lib
    from mass_amd.episodes import room_trajectory
    if fmt is None:
        monkeypatch.delenv("MF_FORMAT", raising=False)
    else:
        monkeypatch.setenv("MF_FORMAT", fmt)
    H, W, M, C, n = 60, 80, 64, 9, 12
    lay, ref = layers(device, "label", C, H, W, M, 0.1)
    if scene == "unrelated":
        fr = sparse_frames(n, H, W, C, seed=41, dmax=3.0)
    else:
        tr = room_trajectory(n, H, W, seed=2, num_classes=C)
        fr = {k: tr[k] for k in ("position", "yaw", "elevation", "depth", "semantic")}
    run_both(lay, ref, fr, slice(0, n), "label", C)
    mode = last_fuse_mode(lay, n)
    want = {"contributions": _lib.MODE_CELLS, "records": _lib.MODE_DENSE, "aggregated": _lib.MODE_CELLS_AGG,
            None: _lib.MODE_CELLS if scene == "unrelated" else _lib.MODE_CELLS_AGG}[fmt]
    assert mode == want
    assert_map_close(lay.data.cpu().numpy(), ref.data.numpy(), what=f"{scene} frames as {fmt or 'the probe chose'}")


@pytest.mark.parametrize("C,my,mx", [(24, 64, 64), (60, 64, 64), (54, 30, 34), (3, 22, 26)])
def test_other_channel_counts_and_maps_that_are_no_multiple_of_the_tile(device, monkeypatch, C, my, mx):
    """Every float4-slot instantiation of fuse_cells_kernel (C <= 8, 16, 32, 56, 64) and maps whose height / width are
    no multiples of four (the rows of a border tile that lie outside the map are neither read nor written).  The small
    map concentrates the points: MF_FORMAT keeps the call with the cells kernel, which this is about."""
    monkeypatch.setenv("MF_FORMAT", "contributions")
    from oracle import massref as orc
    from mass_amd import _lib
    from mass_amd.nn.applications.semantic_projection_layer import SemanticProjectionLayer
    H, W, n = 60, 80, 10
    kw = dict(camera_height=H, camera_width=W, map_height=my, map_width=mx, map_depth=32, grid_resolution=0.1, interpolation_weight=0.5)
    lay = SemanticProjectionLayer(feature_size=C, **kw).train().to(device)
    ref = orc.RefProjectionLayer(feature_size=C, **kw)
    g = torch.Generator().manual_seed(100 + C + my)
    init = torch.rand(my, mx, 32, C, generator=g) * (torch.rand(my, mx, 32, 1, generator=g) < 0.3)
    lay.data.copy_(init)
    ref.data.copy_(init)
    fr = sparse_frames(n, H, W, C, seed=C + mx, dmax=2.5, spread=0.2)
    run_both(lay, ref, fr, slice(0, n), "label", C)
    assert last_fuse_mode(lay, n) == _lib.MODE_CELLS
    assert_map_close(lay.data.cpu().numpy(), ref.data.numpy(), what=f"C={C} map {my}x{mx}x32")

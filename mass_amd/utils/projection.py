"""Functional API of the map update, mirroring the reference's
``mass.utils.projection`` (/root/reference/mass/utils/projection.py) name for
name and argument for argument:

    spherical_to_cartesian   projection.py:6-31     host torch ops (O(1) pose math)
    project_camera_rays      projection.py:34-74    host torch ops (once per layer)
    transform_rays           projection.py:77-110   HIP  mf_transform_rays
    bin_rays                 projection.py:113-230  HIP  mf_bin_rays (+ torch compaction)
    update_feature_map       projection.py:233-351  HIP  mf_update_feature_map

The per-pixel and per-voxel work runs in libmassfuse.so; tensors must live on
a HIP device (no CPU fallback).  ``fuse_frames`` is the fused entry the layers
use (transform + bin + update in one pipeline, nothing materialised).
"""
import numpy as np
import ctypes

import torch

from mass_amd import _lib
from mass_amd._lib import lib, check, ptr, require_device, current_stream


def spherical_to_cartesian(yaw, elevation):
    """Unit vector for a yaw / elevation pair (projection.py:29-31)."""
    return torch.stack([torch.cos(yaw) * torch.cos(elevation),
                        torch.sin(yaw) * torch.cos(elevation),
                        torch.sin(elevation)], dim=-1)


def project_camera_rays(image_height, image_width, focal_length_y, focal_length_x,
                        dtype=torch.float32, device='cpu'):
    """Camera-frame ray per pixel, (rx, -ry, -1) (projection.py:67-74)."""
    kwargs = dict(dtype=dtype, device=device)
    y, x = torch.meshgrid(torch.arange(image_height, **kwargs),
                          torch.arange(image_width, **kwargs), indexing='ij')
    rays_y = (y - 0.5 * float(image_height - 1)) / focal_length_y
    rays_x = (x - 0.5 * float(image_width - 1)) / focal_length_x
    return torch.stack([rays_x, -rays_y, -torch.ones_like(rays_x)], dim=-1)


def rotation_matrix(eye_vector, up_vector):
    """R = stack([eye x up, up, -eye], dim=-1) (projection.py:104-105); the
    cross product is taken over the last axis (the reference's dim-less
    torch.cross would pick axis 0 for a batch of exactly three poses)."""
    return torch.stack([torch.linalg.cross(eye_vector, up_vector, dim=-1),
                        up_vector, -eye_vector], dim=-1)


def pack_poses(origin, eye_vector, up_vector):
    """[..., 12] fp32 pose rows (origin[3], R[3][3] row-major) for the C ABI."""
    R = rotation_matrix(eye_vector, up_vector)
    return torch.cat([origin.to(torch.float32), R.reshape(*R.shape[:-2], 9).to(torch.float32)], dim=-1)


def _f32c(t):
    return t.to(torch.float32).contiguous()


def transform_rays(rays, eye_vector, up_vector):
    """Camera rays [..., H, W, 3] -> world rays (projection.py:77-110).

    ``rays`` may carry a leading batch that matches ``eye_vector`` [B, 3]; the
    same camera rays for every pose is the common case and is not expanded."""
    require_device(rays, eye_vector, up_vector)
    batched = eye_vector.dim() == 2
    eye = eye_vector if batched else eye_vector.unsqueeze(0)
    up = up_vector if batched else up_vector.unsqueeze(0)
    n_frames = eye.shape[0]
    poses = _f32c(pack_poses(torch.zeros_like(eye), eye, up))
    stream = current_stream(rays.device)
    if batched and rays.dim() == 4 and rays.shape[0] == n_frames and n_frames > 1:
        outs = [transform_rays(rays[b], eye[b], up[b]) for b in range(n_frames)]
        return torch.stack(outs, dim=0)
    cam = _f32c(rays[0] if (batched and rays.dim() == 4) else rays)
    n_pix = cam.numel() // 3
    out = torch.empty((n_frames,) + tuple(cam.shape), dtype=torch.float32, device=cam.device)
    check(lib.mf_transform_rays(ptr(cam), n_pix, ptr(poses), n_frames, ptr(out), stream))
    return out if batched else out[0]


def bin_rays_dense(bins0, bins1, bins2, origin, rays, depth, min_ray_depth=0.0, max_ray_depth=10.0):
    """Uncompacted bin_rays: (ind0, ind1, ind2, ratio0, ratio1, ratio2, valid)
    shaped like ``depth`` without its last axis."""
    require_device(bins0, bins1, bins2, origin, rays, depth)
    depth = _f32c(depth)
    lead = depth.shape[:-1]
    if origin.dim() == 1:
        n_frames, o = 1, _f32c(origin.unsqueeze(0))
    else:
        n_frames, o = origin.shape[0], _f32c(origin)
        if lead[0] != n_frames:
            raise ValueError("origin batch does not match depth batch")
    n_pix = depth.numel() // n_frames
    per_frame = int(rays.numel() == depth.numel() * 3)
    if not per_frame and rays.numel() != n_pix * 3:
        raise ValueError("rays must have one 3-vector per depth pixel")
    rays = _f32c(rays)
    b0, b1, b2 = _f32c(bins0), _f32c(bins1), _f32c(bins2)
    dev = depth.device
    ind = [torch.empty(lead, dtype=torch.int64, device=dev) for _ in range(3)]
    rat = [torch.empty(lead, dtype=torch.float32, device=dev) for _ in range(3)]
    valid = torch.empty(lead, dtype=torch.uint8, device=dev)
    check(lib.mf_bin_rays(ptr(b0), b0.numel(), ptr(b1), b1.numel(), ptr(b2), b2.numel(),
                          ptr(o), ptr(rays), per_frame, ptr(depth), n_frames, n_pix,
                          float(min_ray_depth), float(max_ray_depth),
                          ptr(ind[0]), ptr(ind[1]), ptr(ind[2]), ptr(rat[0]), ptr(rat[1]), ptr(rat[2]),
                          ptr(valid), current_stream(dev)))
    return (*ind, *rat, valid)


def bin_rays(bins0, bins1, bins2, origin, rays, depth,
             *features, min_ray_depth=0.0, max_ray_depth=10.0):
    """Voxel index / in-voxel ratio of every valid ray end point, compacted in
    row-major pixel order, plus the matching feature rows (projection.py:113-230)."""
    i0, i1, i2, r0, r1, r2, valid = bin_rays_dense(bins0, bins1, bins2, origin, rays, depth,
                                                   min_ray_depth=min_ray_depth,
                                                   max_ray_depth=max_ray_depth)
    indices = torch.nonzero(valid, as_tuple=True)
    return (i0[indices], i1[indices], i2[indices], r0[indices], r1[indices], r2[indices],
            *[features_i[indices] for features_i in features])


class Workspace:
    """Grow-only device scratch for the fuse pipeline (owned by the caller,
    as the C ABI requires)."""

    def __init__(self):
        self.buf = None

    def get(self, nbytes, device):
        """(256-byte aligned pointer, usable bytes >= nbytes).  The buffer carries 256 spare
        bytes for the alignment, so the usable capacity is numel - 256."""
        if self.buf is None or self.buf.numel() - 256 < nbytes or self.buf.device != device:
            self.buf = torch.empty(int(nbytes) + 256, dtype=torch.uint8, device=device)
        base = self.buf.data_ptr()
        return _lib.c_void_p((base + 255) // 256 * 256), self.buf.numel() - 256


_default_workspace = Workspace()


def _grid_struct(feature_map, bins_x=None, bins_y=None, bins_z=None):
    s0, s1, s2, C = feature_map.shape[-4:]
    g = _lib.MfGrid()
    g.size0, g.size1, g.size2, g.channels = s0, s1, s2, C
    if bins_x is not None:
        g.bins_x, g.bins_y, g.bins_z = bins_x.data_ptr(), bins_y.data_ptr(), bins_z.data_ptr()
        g.n_edges_x, g.n_edges_y, g.n_edges_z = bins_x.numel(), bins_y.numel(), bins_z.numel()
    g.map = feature_map.data_ptr()
    return g


def _feature_kind(features, C):
    """(kind, tensor) for a per-point / per-pixel feature tensor."""
    if features is None:
        return _lib.FEAT_ONES, None
    if features.dtype == torch.uint8:
        return _lib.FEAT_LABEL_U8, features.contiguous()
    if features.dtype == torch.int32:
        return _lib.FEAT_LABEL_I32, features.contiguous()
    if features.dtype == torch.int64:
        return _lib.FEAT_LABEL_I64, features.contiguous()
    return _lib.FEAT_DENSE_F32, _f32c(features)


def _check_map(feature_map):
    require_device(feature_map)
    if feature_map.dtype != torch.float32 or not feature_map.is_contiguous():
        raise ValueError("feature_map must be a contiguous float32 tensor")
    if feature_map.dim() == 5 and feature_map.shape[0] == 1:
        feature_map = feature_map[0]
    if feature_map.dim() != 4:
        raise NotImplementedError("a leading batch of maps is not supported (unused by the reference's callers)")
    return feature_map


def update_feature_map(ind0, ind1, ind2, ratio0, ratio1, ratio2,
                       features, feature_map, interpolation_weight=1.0, workspace=None):
    """Trilinear 8-corner normalised-blend splat of ``features`` [n, C] into
    ``feature_map`` [size0, size1, size2, C], in place (projection.py:233-351).

    Extension: an integer ``features`` tensor [n] is taken as class ids and
    stands for one_hot(features, C).float() without materialising it."""
    fm = _check_map(feature_map)
    require_device(ind0, ind1, ind2, ratio0, ratio1, ratio2, features)
    C = fm.shape[-1]
    n = ind0.numel()
    if n == 0:
        return
    kind, feat = _feature_kind(features, C)
    if kind == _lib.FEAT_DENSE_F32:
        feat = feat.reshape(-1, C)
        if feat.shape[0] != n:
            raise ValueError("features must have one row per point")
    elif kind != _lib.FEAT_ONES and feat.numel() != n:
        raise ValueError("label features must have one id per point")
    i0, i1, i2 = (t.reshape(-1).to(torch.int64).contiguous() for t in (ind0, ind1, ind2))
    r0, r1, r2 = (_f32c(t.reshape(-1)) for t in (ratio0, ratio1, ratio2))
    g = _grid_struct(fm)
    ws = workspace or _default_workspace
    need = lib.mf_fuse_workspace_bytes(g, n, 1)
    if need == 0:
        check(_lib.MF_ERR_INVALID)
    wptr, wbytes = ws.get(need, fm.device)
    check(lib.mf_update_feature_map(g, n, ptr(i0), ptr(i1), ptr(i2), ptr(r0), ptr(r1), ptr(r2),
                                    ptr(feat), kind, float(interpolation_weight), wptr, wbytes,
                                    current_stream(fm.device)))


def _frames_call(bins_x, bins_y, bins_z, cam_rays, poses, depth, features, feature_map,
                 min_ray_depth, max_ray_depth, label_status):
    """Argument blocks of the C ABI for a batch of posed frames: (grid struct, frames struct, B, H, W,
    tensors that must stay alive until the call is issued)."""
    fm = _check_map(feature_map)
    require_device(bins_x, bins_y, bins_z, cam_rays, depth, features)
    H, W = cam_rays.shape[0], cam_rays.shape[1]
    depth = _f32c(depth).reshape(-1, H, W)
    B = depth.shape[0]
    poses = _f32c(poses).reshape(-1, 12)
    if poses.shape[0] != B:
        raise ValueError("one pose row per frame expected")
    if not poses.is_cuda and B != 1:            # a single frame's pose may stay on the host (mf_frames.poses_on_host)
        require_device(poses)
    C = fm.shape[-1]
    kind, feat = _feature_kind(features, C)
    fr = _lib.MfFrames()
    fr.height, fr.width = H, W
    cam = _f32c(cam_rays)
    fr.cam_rays = cam.data_ptr()
    fr.min_depth, fr.max_depth = float(min_ray_depth), float(max_ray_depth)
    fr.feat_kind = kind
    fr.label_status = label_status.data_ptr() if label_status is not None else None
    fr.poses_on_host = 0 if poses.is_cuda else 1
    if kind == _lib.FEAT_DENSE_F32:
        feat = feat.reshape(B, -1, feat.shape[-2], C) if feat.dim() >= 3 else feat
        fr.feat_height, fr.feat_width = feat.shape[1], feat.shape[2]
    elif kind != _lib.FEAT_ONES:
        feat = feat.reshape(B, feat.shape[-2], feat.shape[-1])
        fr.feat_height, fr.feat_width = feat.shape[1], feat.shape[2]
    bx, by, bz = _f32c(bins_x), _f32c(bins_y), _f32c(bins_z)
    g = _grid_struct(fm, bx, by, bz)
    return g, fr, fm, B, H, W, (depth, poses, feat, cam, bx, by, bz)


def fuse_frames(bins_x, bins_y, bins_z, cam_rays, poses, depth, features, feature_map,
                interpolation_weight=0.5, sequential=True, min_ray_depth=0.0, max_ray_depth=10.0,
                workspace=None, label_status=None):
    """transform_rays + bin_rays + update_feature_map for a batch of posed
    frames in one fused pipeline (what BaseProjectionLayer.update runs).

    cam_rays [H, W, 3]; poses [B, 12] (pack_poses); depth [B, H, W(, 1)];
    features: None (ones, C == 1), integer class ids [B, h, w] or fp32
    [B, h, w, C] with h | H and w | W.  sequential=True reproduces B successive
    layer.update() calls, False the functional API's merged batch.

    label_status: optional pinned-host (or device) int32 tensor; the kernels set it to 1 and leave
    the map untouched when a pixel carries a class id outside [0, C) (mf_frames.label_status)."""
    g, fr, fm, B, H, W, (depth, poses, feat, *_keep) = _frames_call(
        bins_x, bins_y, bins_z, cam_rays, poses, depth, features, feature_map, min_ray_depth, max_ray_depth, label_status)
    ws = workspace or _default_workspace
    stream = current_stream(fm.device)
    step = _lib.MAX_FRAMES_PER_CALL if sequential else B
    for b0 in range(0, B, step):
        if b0 > 0 and label_status is not None:
            # a batch of more than 256 frames is issued in parts: the earlier part's kernels are waited for before its
            # status word is read (the word is written by the device; unwaited, the check almost never saw it), so that
            # no part behind an offending one is applied.  Costs a wait only for such batches.
            torch.cuda.current_stream(fm.device).synchronize()
            if int(label_status.reshape(-1)[0]) != 0:
                break
        nb = min(step, B - b0)
        fr.n_frames = nb
        fr.poses = poses[b0:].data_ptr()
        fr.depth = depth[b0:].data_ptr()
        fr.feat = feat[b0:].data_ptr() if feat is not None else None
        need = lib.mf_fuse_workspace_bytes(g, nb * H * W, nb if sequential else 1)
        if need == 0:
            check(_lib.MF_ERR_INVALID)
        wptr, wbytes = ws.get(need, fm.device)
        check(lib.mf_fuse_frames(g, fr, float(interpolation_weight),
                                 _lib.MODE_SEQUENTIAL if sequential else _lib.MODE_MERGED,
                                 wptr, wbytes, stream))


def fuse_frame_maps(updates, sequential=True, min_ray_depth=0.0, max_ray_depth=10.0):
    """One batch of posed frames onto several maps in one library call (mf_fuse_frame_maps): what the
    reference's agent does with a loop of layer.update(observations) per simulator step
    (navigation_policy.py:164-171).  `updates` is a list of dicts, one per map, with the arguments of
    fuse_frames: bins_x, bins_y, bins_z, cam_rays, poses, depth, features, feature_map,
    interpolation_weight, workspace, label_status.  The maps share their voxel grid (equal edges: the
    caller's responsibility) and the frames: cam_rays, poses and depth of updates[0] are used for all.
    The result on every map equals its own fuse_frames call; a single frame (or a merged batch) is
    bucketed once and the maps' tile kernels run side by side."""
    n = len(updates)
    if n < 1 or n > _lib.MAX_MAPS_PER_CALL:
        raise ValueError(f"1 to {_lib.MAX_MAPS_PER_CALL} maps per call, got {n}")
    u0 = updates[0]
    # one tensor each for rays, poses and depth: every map's argument block points at the same memory
    cam = _f32c(u0["cam_rays"])
    H, W = cam.shape[0], cam.shape[1]
    depth = _f32c(u0["depth"]).reshape(-1, H, W)
    poses = _f32c(u0["poses"]).reshape(-1, 12)
    B = depth.shape[0]
    if poses.shape[0] != B:
        raise ValueError("one pose row per frame expected")
    if sequential and B > _lib.MAX_FRAMES_PER_CALL:
        raise ValueError(f"at most {_lib.MAX_FRAMES_PER_CALL} sequential frames per call")
    require_device(cam, depth)
    if not poses.is_cuda and B != 1:            # a single frame's pose may stay on the host (mf_frames.poses_on_host)
        require_device(poses)
    grids, frames = (_lib.MfGrid * n)(), (_lib.MfFrames * n)()
    weights, wptrs, wbytes = (_lib.c_float * n)(), (_lib.c_void_p * n)(), (_lib.c_size_t * n)()
    keep, spaces = [], []
    # what all maps share of the frames block
    fr = frames[0]
    fr.struct_size = ctypes.sizeof(_lib.MfFrames)
    fr.n_frames, fr.height, fr.width = B, H, W
    fr.cam_rays, fr.poses, fr.depth = cam.data_ptr(), poses.data_ptr(), depth.data_ptr()
    fr.min_depth, fr.max_depth = float(min_ray_depth), float(max_ray_depth)
    fr.poses_on_host = 0 if poses.is_cuda else 1
    G = B if sequential else 1
    for m, u in enumerate(updates):
        fm, bx, by, bz = u["feature_map"], u["bins_x"], u["bins_y"], u["bins_z"]
        # the grid block of a map is rebuilt (and its tensors looked at) only when one of its tensors is another one
        gkey = (fm.data_ptr(), bx.data_ptr(), by.data_ptr(), bz.data_ptr(), fm.shape, fm.dtype, bx.dtype, by.dtype, bz.dtype)
        g = _GRID_BLOCKS.get(gkey)
        if g is None:
            fm4 = _check_map(fm)
            if not (bx.dtype == by.dtype == bz.dtype == torch.float32 and bx.is_contiguous() and by.is_contiguous() and bz.is_contiguous()):
                raise ValueError("bin edges must be contiguous float32 tensors")
            require_device(bx, by, bz)
            g = _grid_struct(fm4, bx, by, bz)
            if len(_GRID_BLOCKS) > 64:
                _GRID_BLOCKS.clear()
            _GRID_BLOCKS[gkey] = g
        grids[m] = g
        C = g.channels
        kind, feat = _feature_kind(u.get("features"), C)
        if m > 0:
            frames[m] = frames[0]
        fr = frames[m]
        fr.feat_kind = kind
        status = u.get("label_status")
        fr.label_status = status.data_ptr() if status is not None else None
        fr.feat, fr.feat_height, fr.feat_width = None, 0, 0
        if feat is not None:
            require_device(feat)
            if kind == _lib.FEAT_DENSE_F32:
                feat = feat.reshape(B, -1, feat.shape[-2], C) if feat.dim() >= 3 else feat
            else:
                feat = feat.reshape(B, feat.shape[-2], feat.shape[-1])
            fr.feat_height, fr.feat_width = feat.shape[1], feat.shape[2]
            fr.feat = feat.data_ptr()
        wkey = (g.size0, g.size1, g.size2, C, B * H * W, G)
        need = _WORKSPACE_BYTES.get(wkey)
        if need is None:
            need = lib.mf_fuse_workspace_bytes(g, B * H * W, G)
            if need == 0:
                check(_lib.MF_ERR_INVALID)
            if len(_WORKSPACE_BYTES) > 256:
                _WORKSPACE_BYTES.clear()
            _WORKSPACE_BYTES[wkey] = need
        ws = u.get("workspace")
        if ws is None or any(ws is w for w in spaces):
            raise ValueError("every map of a fuse_frame_maps call needs a Workspace of its own")
        spaces.append(ws)
        wp, wb = ws.get(need, fm.device)
        weights[m] = float(u.get("interpolation_weight", 0.5))
        wptrs[m], wbytes[m] = wp.value, wb
        keep.append((feat, fm, bx, by, bz))
    check(lib.mf_fuse_frame_maps(grids, frames, weights, n, _lib.MODE_SEQUENTIAL if sequential else _lib.MODE_MERGED,
                                 wptrs, wbytes, current_stream(u0["feature_map"].device)))


_WORKSPACE_BYTES = {}      # (map shape, points, groups) -> mf_fuse_workspace_bytes (a pure function of them)
_GRID_BLOCKS = {}          # (pointers, shape, dtypes of a map and its edges) -> checked mf_grid block


class FusePipeline:
    """Batch after batch into one map with the bucketing of batch k+1 overlapped with the tile
    kernels of batch k (mf_fuse_frames_stage on a side stream, mf_fuse_frames_commit in order on the
    caller's stream; two workspaces).  Same results as calling fuse_frames per batch.

        pipe = FusePipeline(device)
        for batch in batches: pipe.submit(bins_x, ..., feature_map, ...)    # at most 256 frames each
        pipe.flush()                                                         # commits the last batch
    """

    def __init__(self, device):
        self.device = torch.device(device)
        self.side = torch.cuda.Stream(self.device)
        self.ws = [Workspace(), Workspace()]
        self.staged = [torch.cuda.Event(), torch.cuda.Event()]
        self.committed = [None, None]
        self.pending = None          # (slot, grid, frames, iw, mode, wptr, wbytes, keep-alive tensors)
        self.k = 0

    def submit(self, bins_x, bins_y, bins_z, cam_rays, poses, depth, features, feature_map,
               interpolation_weight=0.5, sequential=True, min_ray_depth=0.0, max_ray_depth=10.0, label_status=None):
        g, fr, fm, B, H, W, keep = _frames_call(bins_x, bins_y, bins_z, cam_rays, poses, depth, features, feature_map,
                                                min_ray_depth, max_ray_depth, label_status)
        if sequential and B > _lib.MAX_FRAMES_PER_CALL:
            raise ValueError(f"at most {_lib.MAX_FRAMES_PER_CALL} sequential frames per submitted batch")
        depth_t, poses_t, feat_t = keep[0], keep[1], keep[2]
        fr.n_frames, fr.poses, fr.depth = B, poses_t.data_ptr(), depth_t.data_ptr()
        fr.feat = feat_t.data_ptr() if feat_t is not None else None
        mode = _lib.MODE_SEQUENTIAL if sequential else _lib.MODE_MERGED
        need = lib.mf_fuse_workspace_bytes(g, B * H * W, B if sequential else 1)
        if need == 0:
            check(_lib.MF_ERR_INVALID)
        slot = self.k & 1
        main = torch.cuda.current_stream(self.device)
        inputs = torch.cuda.Event()
        inputs.record(main)                               # the inputs were produced on the caller's stream
        # The batch staged by the previous submit is committed FIRST: the staging call below waits on the host for its
        # probe (a few words read back: which entry format the frames ask for), i.e. for everything the side stream waits
        # for, and the caller's stream must have its work queued by then.
        self._commit()
        self.side.wait_event(inputs)
        if self.committed[slot] is not None:
            self.side.wait_event(self.committed[slot])    # the workspace is free once its last commit is done
        with torch.cuda.stream(self.side):
            wptr, wbytes = self.ws[slot].get(need, self.device)
            check(lib.mf_fuse_frames_stage(g, fr, float(interpolation_weight), mode, wptr, wbytes,
                                           _lib.c_void_p(self.side.cuda_stream)))
            self.staged[slot].record(self.side)
        self.pending = (slot, g, fr, float(interpolation_weight), mode, wptr, wbytes, keep, fm)
        self.k += 1

    def _commit(self):
        if self.pending is None:
            return
        slot, g, fr, iw, mode, wptr, wbytes, _keep, fm = self.pending
        main = torch.cuda.current_stream(self.device)
        main.wait_event(self.staged[slot])
        check(lib.mf_fuse_frames_commit(g, fr, iw, mode, wptr, wbytes, _lib.c_void_p(main.cuda_stream)))
        ev = torch.cuda.Event()
        ev.record(main)
        self.committed[slot] = ev
        self.pending = None

    def flush(self):
        self._commit()


def unproject_bin(bins_x, bins_y, bins_z, cam_rays, poses, depth, min_ray_depth=0.0, max_ray_depth=10.0):
    """Fused transform_rays + bin_rays (uncompacted), in the (x, y, z) order of
    bin_rays as BaseProjectionLayer.update calls it; parity/debug entry."""
    require_device(bins_x, bins_y, bins_z, cam_rays, depth)
    H, W = cam_rays.shape[0], cam_rays.shape[1]
    depth = _f32c(depth).reshape(-1, H, W)
    B = depth.shape[0]
    poses = _f32c(poses).reshape(B, 12)
    if not poses.is_cuda and B != 1:            # a single frame's pose may stay on the host (mf_frames.poses_on_host)
        require_device(poses)
    cam = _f32c(cam_rays)
    bx, by, bz = _f32c(bins_x), _f32c(bins_y), _f32c(bins_z)
    dummy = torch.empty(1, dtype=torch.float32, device=depth.device)
    g = _lib.MfGrid()
    g.size0, g.size1, g.size2, g.channels = by.numel() - 1, bx.numel() - 1, bz.numel() - 1, 1
    g.bins_x, g.bins_y, g.bins_z = bx.data_ptr(), by.data_ptr(), bz.data_ptr()
    g.n_edges_x, g.n_edges_y, g.n_edges_z = bx.numel(), by.numel(), bz.numel()
    g.map = dummy.data_ptr()
    fr = _lib.MfFrames()
    fr.n_frames, fr.height, fr.width = B, H, W
    fr.cam_rays, fr.poses, fr.depth = cam.data_ptr(), poses.data_ptr(), depth.data_ptr()
    fr.poses_on_host = 0 if poses.is_cuda else 1
    fr.min_depth, fr.max_depth = float(min_ray_depth), float(max_ray_depth)
    dev = depth.device
    ind = [torch.empty((B, H, W), dtype=torch.int64, device=dev) for _ in range(3)]
    rat = [torch.empty((B, H, W), dtype=torch.float32, device=dev) for _ in range(3)]
    valid = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
    check(lib.mf_unproject_bin(g, fr, ptr(ind[0]), ptr(ind[1]), ptr(ind[2]), ptr(rat[0]), ptr(rat[1]),
                               ptr(rat[2]), ptr(valid), current_stream(dev)))
    return (*ind, *rat, valid)

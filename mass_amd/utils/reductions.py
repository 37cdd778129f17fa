"""Whole-map reductions the reference's callers run on the map tensors,
as single-pass HIP kernels:

    navigable_area   /root/reference/mass/navigation_policy.py:208-221 (body; the method itself
                     belongs to NavigationPolicy, which is out of scope)
    amax_z           /root/reference/agent.py:330-331, 391-392   data.amax(dim=2)
    map_stats        occupied voxels and sum |map| of a map in one pass (the counters of bench.py's episode workload)
"""
import torch
import torch.nn.functional as functional

from mass_amd._lib import lib, check, ptr, require_device, current_stream, MAP_STATS_PARTS


def _map4(data):
    require_device(data)
    if data.dim() != 4 or data.dtype != torch.float32 or not data.is_contiguous():
        raise ValueError("expected a contiguous float32 [H, W, D, C] map")
    return data


def column_occupied(data, depth_slice=None, obstacle_threshold=0.0):
    """[H, W] bool: any voxel of the column slice has an L1 feature norm above the threshold."""
    data = _map4(data)
    H, W, D, C = data.shape
    z0, z1, step = (depth_slice or slice(None)).indices(D)
    if step != 1:
        raise NotImplementedError("depth_slice must be contiguous")
    out = torch.empty(H, W, dtype=torch.uint8, device=data.device)
    check(lib.mf_column_occupied(ptr(data), H, W, D, C, z0, max(z1, z0), float(obstacle_threshold), ptr(out),
                                 current_stream(data.device)))
    return out.bool()


def navigable_area(data, padding=3, depth_slice=None, obstacle_threshold=0.0):
    """1 where the agent can stand: no occupied voxel in the column slice, with `padding`
    voxels of clearance (navigation_policy.py:208-221)."""
    navigable = torch.logical_not(column_occupied(data, depth_slice, obstacle_threshold)).to(dtype=data.dtype)
    return 1 - functional.max_pool2d(1 - navigable.unsqueeze(0), 2 * padding + 1, stride=1,
                                     padding=padding).squeeze(0)


def amax_z(data):
    """data.amax(dim=2): [H, W, C] channel-wise maximum over the depth axis."""
    data = _map4(data)
    H, W, D, C = data.shape
    out = torch.empty(H, W, C, dtype=torch.float32, device=data.device)
    check(lib.mf_amax_z(ptr(data), H, W, D, C, ptr(out), current_stream(data.device)))
    return out


def map_stats(data):
    """(occupied voxels, sum |data|): `int((data != 0).any(-1).sum())` and `float(data.abs().sum())` in one pass over the
    map.  The sum is an exact integer sum of the terms truncated to 2^-24 (the same bits whatever the order).  Waits for
    the stream (16 bytes are read back)."""
    data = _map4(data)
    H, W, D, C = data.shape
    out = torch.empty(2, dtype=torch.int64, device=data.device)
    scratch = torch.empty(2 * MAP_STATS_PARTS, dtype=torch.int64, device=data.device)
    check(lib.mf_map_stats(ptr(data), H, W, D, C, ptr(out), ptr(scratch), current_stream(data.device)))
    occupied, fx = out.tolist()
    return int(occupied), fx / 16777216.0

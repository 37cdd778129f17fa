"""Inter-map instance matching, mirroring the matching block of the reference's
``predict_scene_differences`` (/root/reference/mass/utils/experimentation.py:169-310):

    pairwise_distance        :261-265 (features) and :277-280 (3-d goals)   HIP mf_pairwise_distance
    linear_sum_assignment    :284-287 scipy.optimize.linear_sum_assignment  host C++ mf_linear_sum_assignment
    match_instances          the two together, as :261-287 chains them

The simulator glue of that module (restart loop, ground-truth diffs) is out of scope.
"""
import numpy as np
import torch

from mass_amd import _lib
from mass_amd._lib import lib, check, ptr, require_device, current_stream

_METRICS = {"l2": _lib.METRIC_L2, "l2_gemm": _lib.METRIC_L2_GEMM, "cosine": _lib.METRIC_COSINE}


def pairwise_distance(feature0, feature1, metric="l2"):
    """[N0, D], [N1, D] -> [N0, N1] fp32: torch.linalg.norm(f0[:, None] - f1[None], dim=2)
    for metric "l2" (the reference's arithmetic), the norm-expansion GEMM on the
    fp32 matrix cores for "l2_gemm", 1 - cosine similarity for "cosine"."""
    require_device(feature0, feature1)
    if feature0.dim() != 2 or feature1.dim() != 2 or feature0.shape[1] != feature1.shape[1]:
        raise ValueError("expected [N0, D] and [N1, D]")
    f0 = feature0.to(torch.float32).contiguous()
    f1 = feature1.to(torch.float32).contiguous()
    out = torch.empty(f0.shape[0], f1.shape[0], dtype=torch.float32, device=f0.device)
    if out.numel():
        check(lib.mf_pairwise_distance(ptr(f0), f0.shape[0], ptr(f1), f1.shape[0], f0.shape[1], ptr(out),
                                       _METRICS[metric], current_stream(f0.device)))
    return out


def linear_sum_assignment(cost):
    """Minimum-cost bipartite matching of a [N0, N1] cost (tensor or array);
    returns (row_ind, col_ind) int64 arrays like scipy's function."""
    if isinstance(cost, torch.Tensor):
        cost = cost.detach().cpu().numpy()
    cost = np.ascontiguousarray(cost, dtype=np.float64)
    if cost.ndim != 2:
        raise ValueError("expected a matrix (2-D array), got a %d array" % cost.ndim)
    k = min(cost.shape)
    rows, cols = np.empty(k, np.int64), np.empty(k, np.int64)
    n = check(lib.mf_linear_sum_assignment(cost.ctypes.data, cost.shape[0], cost.shape[1],
                                           rows.ctypes.data, cols.ctypes.data))
    return rows[:n], cols[:n]


def match_instances(feature0, feature1, metric="l2"):
    """experimentation.py:261-287: pairwise cost on the device, assignment on
    the host.  Returns (cost tensor, instance_ids0, instance_ids1)."""
    cost = pairwise_distance(feature0, feature1, metric=metric)
    rows, cols = linear_sum_assignment(cost)
    return cost, rows, cols

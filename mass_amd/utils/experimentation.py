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


def predict_scene_differences(semantic_projection_layer0, semantic_projection_layer1,
                              resnet_projection_layer0, resnet_projection_layer1,
                              objects_moved, object_ids_to_move_pred,
                              confidence_threshold: float = 0.2, contour_padding: int = 3,
                              contour_threshold: float = 0.0, distance_threshold: float = 0.0,
                              deformation_threshold: float = 0.0,
                              id_to_pickable=None, id_to_openable=None):
    """Which object class differs between the walkthrough map (layer0) and the unshuffle map
    (layer1), and where its instances are in each (experimentation.py:169-310).

    Same control flow as the reference: per candidate class, find() in both maps, pairwise
    cost (feature L2 if feature maps are given, else |size0 - size1|; the 3-d goal distance for
    classes that are only openable), Hungarian assignment, then the distance / openable
    criteria.  The reference takes the pickable / openable tables from the external
    `rearrange` package (ID_TO_PICKABLE, ID_TO_OPENABLE); here they are arguments (default:
    every class pickable, none openable)."""
    object_to_move = None
    object_goals0, object_goals1 = [], []
    for candidate_object in object_ids_to_move_pred:
        object_pickable = True if id_to_pickable is None else bool(id_to_pickable[candidate_object])
        object_openable = False if id_to_openable is None else bool(id_to_openable[candidate_object])
        if candidate_object in objects_moved or not any([object_pickable, object_openable]):
            continue
        kwargs = dict(contour_padding=contour_padding, contour_threshold=contour_threshold,
                      confidence_threshold=confidence_threshold)
        conf0, goal0, size0, feature0 = semantic_projection_layer0.find(
            candidate_object, feature_map=resnet_projection_layer0, **kwargs)
        conf1, goal1, size1, feature1 = semantic_projection_layer1.find(
            candidate_object, feature_map=resnet_projection_layer1, **kwargs)
        if len(conf0) == 0 or len(conf1) == 0:
            continue
        if feature0 is not None and feature1 is not None:
            deformation = pairwise_distance(torch.stack(feature0, dim=0), torch.stack(feature1, dim=0))
        else:
            size0, size1 = torch.stack(size0, dim=0), torch.stack(size1, dim=0)
            deformation = (size0.unsqueeze(1) - size1.unsqueeze(0)).abs()
        goal0, goal1 = torch.stack(goal0, dim=0), torch.stack(goal1, dim=0)
        distance = pairwise_distance(goal0, goal1)
        instance_ids0, instance_ids1 = linear_sum_assignment(deformation if object_pickable else distance)
        distance_host = distance.cpu()
        for instance0, instance1 in zip(instance_ids0, instance_ids1):
            instance_move = object_pickable and bool(distance_host[instance0, instance1] > distance_threshold)
            instance_open = object_openable
            if instance_move or instance_open:
                object_to_move = candidate_object
                object_goals0.append(goal0[instance0])
                object_goals1.append(goal1[instance1])
        if object_to_move is not None:
            break
    return object_to_move, object_goals0, object_goals1

"""One simulator step onto several maps.

The reference's agent keeps a dict of projection layers and, per step, hands the SAME posed
observation to each of them in a Python loop (`/root/reference/mass/navigation_policy.py:131-171`,
`update_feature_maps`: `for name in update_map: self.feature_maps[name].update(observations)`;
the maps are built at `agent.py:107-111`).  A single frame is a chain of short dependent kernels
(bucket -> scan -> scatter -> tile kernel) that leaves most of an MI355X idle, and the first half
of the chain - unprojecting and bucketing the frame's points - does not depend on the features at
all.  `update_feature_maps` here hands the layers that share their voxel grid and camera to
`mf_fuse_frame_maps` (include/massfuse.h): the frame is bucketed once and the maps' tile kernels
run side by side.  Every map ends up with the bits its own `layer.update()` would have given it.
"""
from typing import Any, Dict, List, Mapping, Optional, Sequence, Union

import numpy as np
import torch

from mass_amd import _lib
from mass_amd.nn.base_projection_layer import BaseProjectionLayer
from mass_amd.nn.applications.occupancy_projection_layer import OccupancyProjectionLayer
from mass_amd.nn.applications.semantic_projection_layer import SemanticProjectionLayer
from mass_amd.utils.projection import fuse_frame_maps

_SAME_GEOMETRY: Dict[Any, bool] = {}


def _select(feature_maps, update_map) -> List[Any]:
    if isinstance(feature_maps, Mapping):
        if update_map is None:
            return list(feature_maps.values())
        names = [update_map] if not isinstance(update_map, (list, tuple)) else list(update_map)
        return [feature_maps[name] for name in names]       # KeyError on an unknown map, like the reference's dict
    if update_map is not None:
        raise TypeError("update_map names need feature_maps to be a dict of layers")
    return list(feature_maps)


_PLAIN_UPDATES = (BaseProjectionLayer.update, OccupancyProjectionLayer.update, SemanticProjectionLayer.update)


def _plain_update(lay) -> bool:
    """The layer's update() is one of the three this module knows how to restate (a subclass that
    overrides update(), like ResNetProjectionLayer with its subsampled depth, keeps its own call)."""
    return type(lay).update in _PLAIN_UPDATES


def _geometry_stamp(lay):
    """Identity and version of the tensors that fix where a pixel lands: camera rays and voxel edges."""
    b = lay._buffers
    r, x, y, z = b["rays"], b["bins_x"], b["bins_y"], b["bins_z"]
    return (r.data_ptr(), r._version, x.data_ptr(), x._version, y.data_ptr(), y._version, z.data_ptr(), z._version)


def _same_geometry(a, b, stamp_a=None) -> bool:
    """Equal voxel edges and camera rays, by value (looked at once per version of the tensors: reset()
    rewrites the edges)."""
    key = (stamp_a or _geometry_stamp(a), _geometry_stamp(b))
    hit = _SAME_GEOMETRY.get(key)
    if hit is None:
        ta, tb = [(l.rays, l.bins_x, l.bins_y, l.bins_z) for l in (a, b)]
        hit = (a.data.shape[:3] == b.data.shape[:3] and a.data.device == b.data.device and
               all(x.shape == y.shape and x.dtype == y.dtype and bool(torch.equal(x, y)) for x, y in zip(ta, tb)))
        if len(_SAME_GEOMETRY) > 256:
            _SAME_GEOMETRY.clear()
        _SAME_GEOMETRY[key] = hit
    return hit


def update_feature_maps(feature_maps: Union[Mapping[str, Any], Sequence[Any]],
                        observations: Dict[str, Any],
                        update_map: Optional[Union[str, List[str]]] = None,
                        validate=True, shared: bool = True):
    """`layer.update(observations)` for every selected layer (navigation_policy.py:164-171).

    feature_maps: dict name -> projection layer (the reference's `self.feature_maps`) or a sequence
    of layers; update_map: a name or list of names (None: all of them, in dict order).
    validate: the class-id check of the semantic layers (SemanticProjectionLayer.update): True raises
    from this call (one wait at the end, for all layers), "defer" at the next call into the layer,
    False skips it.  shared=False is the plain loop.

    Every layer reads what it needs from the one observation dict (`features`, `semantic`, nothing
    for an occupancy map); host arrays are uploaded once, not once per layer."""
    layers = _select(feature_maps, update_map)
    if not layers:
        return
    sem_validate = "defer" if validate is True else validate

    group: List[Any] = []
    if shared:
        lead_stamp = None
        for lay in layers:
            if (len(group) < _lib.MAX_MAPS_PER_CALL and isinstance(lay, BaseProjectionLayer) and _plain_update(lay) and
                    lay.data.is_cuda and all(lay is not g for g in group) and
                    (not group or _same_geometry(group[0], lay, lead_stamp))):
                group.append(lay)
                if lead_stamp is None:
                    lead_stamp = _geometry_stamp(lay)
    if len(group) < 2:
        group = []

    if group:
        lead = group[0]
        dev = lead.data.device
        uploaded = {}

        def on_device(key):           # host arrays: one upload for all maps
            value = observations[key]
            if isinstance(value, torch.Tensor) and value.device == dev:
                return value
            if key not in uploaded:
                if isinstance(value, np.ndarray) and not value.flags.writeable:
                    value = value.copy()
                uploaded[key] = torch.as_tensor(value).to(dev, non_blocking=True)
            return uploaded[key]

        depth = on_device("depth")
        poses = lead._poses(observations["position"], observations["yaw"], observations["elevation"], cache=False)
        updates = []
        for lay in group:
            features, status = None, None
            if isinstance(lay, SemanticProjectionLayer):
                if sem_validate:
                    lay.check_labels(synchronize=False)
                features = lay._labels(on_device("semantic"))
                status = lay._status() if sem_validate else None
            elif not isinstance(lay, OccupancyProjectionLayer):
                features = on_device("features")
                if features.dtype != lay.data.dtype:
                    features = features.to(lay.data.dtype)
            lay._map_version += 1
            b = lay._buffers
            updates.append(dict(bins_x=b["bins_x"], bins_y=b["bins_y"], bins_z=b["bins_z"], features=features,
                                feature_map=lay.data, interpolation_weight=lay.interpolation_weight,
                                workspace=lay._workspace, label_status=status))
        updates[0].update(cam_rays=lead._buffers["rays"], poses=poses, depth=depth)
        fuse_frame_maps(updates, sequential=True)
    for lay in layers:
        if not any(lay is g for g in group):
            if isinstance(lay, SemanticProjectionLayer):
                lay.update(observations, validate=sem_validate)
            else:
                lay.update(observations)
    if validate is True:
        sem = [lay for lay in layers if isinstance(lay, SemanticProjectionLayer)]
        if sem:
            torch.cuda.current_stream(sem[0].data.device).synchronize()
            # every semantic layer is looked at (and its status word cleared) before anything is raised: a layer left with
            # its word set would raise a stale error from its NEXT update and drop a valid observation.  Unlike the
            # reference's loop, which stops at the first bad layer, the other maps of the step have been updated by then.
            failure = None
            for lay in sem:
                try:
                    lay.check_labels(synchronize=False)
                except RuntimeError as err:
                    failure = failure or err
            if failure is not None:
                raise failure

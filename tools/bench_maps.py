#!/usr/bin/env python3
"""One simulator step onto three maps (BASELINE.json configs[2]: occupancy C=1, semantic C=54 class ids, RGB C=3,
256^3 each, 480x640 frames of the box-room trajectory): the reference's loop of layer.update() calls
(navigation_policy.py:164-171) beside mass_amd.nn.update_feature_maps (one mf_fuse_frame_maps call).
Prints one JSON line per mode: wall time per frame, host issue time per frame, host time inside the C call."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mass_amd.episodes import room_trajectory
from mass_amd.nn.base_projection_layer import BaseProjectionLayer
from mass_amd.nn.applications.occupancy_projection_layer import OccupancyProjectionLayer
from mass_amd.nn.applications.semantic_projection_layer import SemanticProjectionLayer
from mass_amd.nn.feature_maps import update_feature_maps
from mass_amd.utils import projection as _pj

H, W, M = 480, 640, 256
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
modes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["loop", "shared"]
kw = dict(camera_height=H, camera_width=W, map_height=M, map_width=M, map_depth=M, grid_resolution=0.05)
tr = room_trajectory(n, H, W, seed=1)
maps = dict(occupancy=OccupancyProjectionLayer(**kw).to(dev), semantic=SemanticProjectionLayer(feature_size=54, **kw).to(dev),
            rgb=BaseProjectionLayer(feature_size=3, **kw).to(dev))
d, s, c = tr["depth"].to(dev), tr["semantic"].to(dev)[..., None], tr["rgb"].to(dev)
obs = [dict(position=tr["position"][t], yaw=tr["yaw"][t], elevation=tr["elevation"][t], depth=d[t], semantic=s[t], features=c[t])
       for t in range(n)]

_real, _acc = _pj.lib.mf_fuse_frame_maps, [0.0, 0]


def _timed(*a):
    t0 = time.perf_counter()
    r = _real(*a)
    _acc[0] += time.perf_counter() - t0
    _acc[1] += 1
    return r


_pj.lib.mf_fuse_frame_maps = _timed
for mode in modes:
    best = None
    for rep in range(3):
        for lay in maps.values():
            lay.reset()
        _acc[0], _acc[1] = 0.0, 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for o in obs:
            if mode == "loop":
                maps["occupancy"].update(o)
                maps["semantic"].update(o, validate="defer")
                maps["rgb"].update(o)
            else:
                update_feature_maps(maps, o, validate="defer")
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        maps["semantic"].check_labels()
        row = dict(mode=mode, frames=n, ms_per_frame=(t2 - t0) / n * 1e3, frames_per_s=n / (t2 - t0),
                   host_issue_ms_per_frame=(t1 - t0) / n * 1e3,
                   host_ms_in_mf_fuse_frame_maps=_acc[0] / max(_acc[1], 1) * 1e3 if mode != "loop" else None)
        if rep > 0 and (best is None or row["ms_per_frame"] < best["ms_per_frame"]):
            best = row
    print(json.dumps(best), flush=True)

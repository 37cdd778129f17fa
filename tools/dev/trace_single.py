#!/usr/bin/env python3
"""dev: config-3 style per-frame updates of three maps (room trajectory) for a rocprofv3 kernel trace."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mass_amd.episodes import room_trajectory
from mass_amd.nn.base_projection_layer import BaseProjectionLayer
from mass_amd.nn.applications.occupancy_projection_layer import OccupancyProjectionLayer
from mass_amd.nn.applications.semantic_projection_layer import SemanticProjectionLayer
H, W, M = 480, 640, 256
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
kw = dict(camera_height=H, camera_width=W, map_height=M, map_width=M, map_depth=M, grid_resolution=0.05)
tr = room_trajectory(n, H, W, seed=1)
occ = OccupancyProjectionLayer(**kw).to(dev)
sem = SemanticProjectionLayer(feature_size=54, **kw).to(dev)
rgb = BaseProjectionLayer(feature_size=3, **kw).to(dev)
d, s, c = tr["depth"].to(dev), tr["semantic"].to(dev)[..., None], tr["rgb"].to(dev)
from mass_amd.nn.feature_maps import update_feature_maps
from mass_amd.utils import projection as _pj
_real = _pj.lib.mf_fuse_frame_maps
_acc = [0.0, 0]
def _timed(*a):
    t = time.perf_counter(); r = _real(*a); _acc[0] += time.perf_counter() - t; _acc[1] += 1; return r
if os.environ.get("TIME_C"):
    _pj.lib.mf_fuse_frame_maps = _timed
maps = dict(occupancy=occ, semantic=sem, rgb=rgb)
mode = sys.argv[2] if len(sys.argv) > 2 else "loop"
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(n):
        o = dict(position=tr["position"][t], yaw=tr["yaw"][t], elevation=tr["elevation"][t], depth=d[t])
        if mode == "loop":
            occ.update(o)
            sem.update(dict(o, semantic=s[t]), validate="defer")
            rgb.update(dict(o, features=c[t]))
        else:
            update_feature_maps(maps, dict(o, semantic=s[t], features=c[t]), validate="defer", shared=mode == "shared")
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print("ms per frame (3 maps)", (time.perf_counter() - t0) / n * 1e3, "issue only", (t1 - t0) / n * 1e3,
          "C call us", _acc[0] / max(_acc[1], 1) * 1e6, flush=True)
    _acc[0] = 0.0; _acc[1] = 0

if mode != 'loop':
    sys.exit(0)
# host cost of one update per layer with the GPU idle between calls
for name, lay, extra in (("occ", occ, {}), ("sem", sem, {"semantic": s[0]}), ("rgb", rgb, {"features": c[0]})):
    ts = []
    for t in range(n):
        o = dict(position=tr["position"][t], yaw=tr["yaw"][t], elevation=tr["elevation"][t], depth=d[t], **extra)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        lay.update(o, validate="defer") if name == "sem" else lay.update(o)
        ts.append(time.perf_counter() - t0)
    ts.sort()
    print(name, "host ms per update (median)", ts[len(ts) // 2] * 1e3, flush=True)

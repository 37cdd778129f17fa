#!/usr/bin/env python3
"""dev: cProfile of update_feature_maps (host side), C call stubbed out to see the Python share."""
import os, sys, cProfile, pstats, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mass_amd.episodes import room_trajectory
from mass_amd.nn.base_projection_layer import BaseProjectionLayer
from mass_amd.nn.applications.occupancy_projection_layer import OccupancyProjectionLayer
from mass_amd.nn.applications.semantic_projection_layer import SemanticProjectionLayer
from mass_amd.nn.feature_maps import update_feature_maps
from mass_amd.utils import projection as _pj
H, W, M = 480, 640, 256
dev = torch.device("cuda:0")
n = 200
kw = dict(camera_height=H, camera_width=W, map_height=M, map_width=M, map_depth=M, grid_resolution=0.05)
tr = room_trajectory(n, H, W, seed=1)
maps = dict(occupancy=OccupancyProjectionLayer(**kw).to(dev), semantic=SemanticProjectionLayer(feature_size=54, **kw).to(dev),
            rgb=BaseProjectionLayer(feature_size=3, **kw).to(dev))
d, s, c = tr["depth"].to(dev), tr["semantic"].to(dev)[..., None], tr["rgb"].to(dev)
obs = [dict(position=tr["position"][t], yaw=tr["yaw"][t], elevation=tr["elevation"][t], depth=d[t], semantic=s[t], features=c[t]) for t in range(n)]
def run():
    for o in obs:
        update_feature_maps(maps, o, validate="defer")
    torch.cuda.synchronize()
run()
real = _pj.lib.mf_fuse_frame_maps
_pj.lib.mf_fuse_frame_maps = lambda *a: 0
t0 = time.perf_counter(); run(); print("python only us/frame", (time.perf_counter() - t0) / n * 1e6)
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)

#!/usr/bin/env python3
"""dev: cProfile of per-frame layer.update() calls (host side)."""
import os, sys, cProfile, pstats
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mass_amd.episodes import room_trajectory
from mass_amd.nn.base_projection_layer import BaseProjectionLayer
from mass_amd.nn.applications.occupancy_projection_layer import OccupancyProjectionLayer
from mass_amd.nn.applications.semantic_projection_layer import SemanticProjectionLayer
H, W, M = 480, 640, 256
dev = torch.device("cuda:0")
n = 100
kw = dict(camera_height=H, camera_width=W, map_height=M, map_width=M, map_depth=M, grid_resolution=0.05)
tr = room_trajectory(n, H, W, seed=1)
occ = OccupancyProjectionLayer(**kw).to(dev)
sem = SemanticProjectionLayer(feature_size=54, **kw).to(dev)
rgb = BaseProjectionLayer(feature_size=3, **kw).to(dev)
d, s, c = tr["depth"].to(dev), tr["semantic"].to(dev)[..., None], tr["rgb"].to(dev)
from mass_amd.nn.feature_maps import update_feature_maps
maps = dict(occupancy=occ, semantic=sem, rgb=rgb)
def run():
    for t in range(n):
        o = dict(position=tr["position"][t], yaw=tr["yaw"][t], elevation=tr["elevation"][t], depth=d[t])
        update_feature_maps(maps, dict(o, semantic=s[t], features=c[t]), validate="defer")
    torch.cuda.synchronize()
run()
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)

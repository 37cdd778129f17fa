#!/usr/bin/env python3
"""Per-frame latency of layer.update() (the way agent.py drives the maps: one
frame per call, three maps per simulator step) — BASELINE.json configs[2]."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mass_amd import _lib
from mass_amd.episodes import dist_a_frames, room_trajectory
from mass_amd.nn.base_projection_layer import BaseProjectionLayer
from mass_amd.nn.applications.occupancy_projection_layer import OccupancyProjectionLayer
from mass_amd.nn.applications.semantic_projection_layer import SemanticProjectionLayer

H, W, M = 480, 640, 256
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
kw = dict(camera_height=H, camera_width=W, map_height=M, map_width=M, map_depth=M, grid_resolution=0.05)
for workload in ("distA", "room"):
    fr = dist_a_frames(n) if workload == "distA" else room_trajectory(n, H, W, seed=0)
    rgb = fr["rgb"] if "rgb" in fr else torch.rand(n, H, W, 3, generator=torch.Generator().manual_seed(0))
    depth = fr["depth"].to(dev)
    sem = fr["semantic"].to(dev)
    rgb = rgb.to(dev)
    layers = dict(occupancy=OccupancyProjectionLayer(**kw).to(dev), semantic=SemanticProjectionLayer(feature_size=54, **kw).to(dev),
                  rgb=BaseProjectionLayer(feature_size=3, **kw).to(dev))
    for name, lay in layers.items():
        def upd(t):
            o = dict(position=fr["position"][t], yaw=fr["yaw"][t], elevation=fr["elevation"][t], depth=depth[t])
            if name == "semantic":
                o["semantic"] = sem[t][..., None]
            if name == "rgb":
                o["features"] = rgb[t]
            lay.update(o)
        for t in range(4):
            upd(t)
        torch.cuda.synchronize()
        _lib.check(_lib.lib.mf_profile_enable(1))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for t in range(4, n):
            upd(t)
        e1.record()
        torch.cuda.synchronize()
        ms = np.zeros((n - 4, 5), np.float32)
        for k in range(n - 4):
            _lib.check(_lib.lib.mf_profile_read(k, ms[k].ctypes.data))
        _lib.check(_lib.lib.mf_profile_enable(0))
        print(json.dumps(dict(workload=workload, map=name, frames=n - 4, ms_per_update_wall=e0.elapsed_time(e1) / (n - 4),
                              gpu_ms=dict(zip(["zero+count", "scan", "scatter", "fuse_tiles", "call"],
                                              [round(float(x), 4) for x in ms.mean(0)])))), flush=True)

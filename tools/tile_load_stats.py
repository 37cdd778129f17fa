#!/usr/bin/env python3
"""Per-tile record counts of one frame (dev): how concentrated a scene is on the map tiles."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mass_amd.episodes import dist_a_frames, room_trajectory
from mass_amd.nn.applications.semantic_projection_layer import SemanticProjectionLayer
from mass_amd.utils.projection import unproject_bin
dev = torch.device("cuda:0")
H, W, M = 480, 640, 256
lay = SemanticProjectionLayer(camera_height=H, camera_width=W, map_height=M, map_width=M, map_depth=M, feature_size=54, grid_resolution=0.05).to(dev)
for name, fr in (("room", room_trajectory(8, H, W, seed=0)), ("distA", dist_a_frames(2))):
    for t in (0, fr["depth"].shape[0] - 1):
        poses = lay._poses(fr["position"][t], fr["yaw"][t], fr["elevation"][t])
        ix, iy, iz, rx, ry, rz, valid = unproject_bin(lay.bins_x, lay.bins_y, lay.bins_z, lay.rays, poses, fr["depth"][t:t + 1].to(dev))
        v = valid.bool()
        for (s0, s1, s2) in ((2, 2, 3), (3, 3, 3)):
            axes = []
            for k, r, size, sh in ((iy[v], ry[v], M, s0), (ix[v], rx[v], M, s1), (iz[v], rz[v], M, s2)):
                lo = torch.where(r < 0.5, (k - 1).clamp(min=0), k) >> sh
                hi = torch.where(r < 0.5, k, (k + 1).clamp(max=size - 1)) >> sh
                axes.append((lo, hi))
            nt1, nt2 = M >> s1, M >> s2
            keys = torch.stack([((a * nt1 + b) * nt2 + c) for a in axes[0] for b in axes[1] for c in axes[2]], 1)
            keys = torch.sort(keys, 1).values
            first = torch.ones_like(keys, dtype=torch.bool); first[:, 1:] = keys[:, 1:] != keys[:, :-1]
            cnt = torch.bincount(keys[first]).cpu().numpy()
            cnt = np.sort(cnt[cnt > 0])[::-1]
            cum = np.cumsum(cnt) / cnt.sum()
            print(name, "frame", t, "tile", (1 << s0, 1 << s1, 1 << s2), "records", int(cnt.sum()), "tiles", len(cnt), "max", int(cnt[0]),
                  "top1/5/20/50 share", [round(float(cum[min(i, len(cum) - 1)]), 2) for i in (0, 4, 19, 49)],
                  "tiles >4096:", int((cnt > 4096).sum()), ">2048:", int((cnt > 2048).sum()), ">1024:", int((cnt > 1024).sum()))

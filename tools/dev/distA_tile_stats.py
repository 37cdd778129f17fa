#!/usr/bin/env python3
"""CPU statistics of the headline batch (64 distribution-A frames -> 256^3) on 4x4x8 tiles (dev):
records, in-tile corner contributions and (voxel, frame) cells per tile.  Approximate geometry (uniform bins)."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mass_amd.episodes import dist_a_frames
from mass_amd.utils.projection import project_camera_rays, spherical_to_cartesian, rotation_matrix
H, W, M, RES = 480, 640, 256, 0.05
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
S = (2, 2, 3)
fr = dist_a_frames(B)
cam = project_camera_rays(H, W, 240.0, 240.0).reshape(-1, 3).numpy()
lo_edge = -(M + 1) * RES / 2
recs_t, cells, contribs = [], [], []
nt = (M >> S[0], M >> S[1], M >> S[2])
rec_cnt = np.zeros(nt[0] * nt[1] * nt[2], np.int64)
con_cnt = np.zeros_like(rec_cnt); cell_cnt = np.zeros_like(rec_cnt); vc_cnt = np.zeros_like(rec_cnt)
vox_touched = np.zeros(M * M * M, bool)
for f in range(B):
    eye = spherical_to_cartesian(fr["yaw"][f], fr["elevation"][f]); up = spherical_to_cartesian(fr["yaw"][f], fr["elevation"][f] + np.pi / 2)
    R = rotation_matrix(eye, up).numpy()
    q = cam @ R.T
    d = fr["depth"][f].reshape(-1, 1).numpy()
    p = fr["position"][f].numpy()[None] + q * d
    t = (p - lo_edge) / RES
    k = np.floor(t).astype(np.int64); r = (t - k).astype(np.float32)
    ok = ((k >= 0) & (k < M)).all(1)
    k, r = k[ok], r[ok]
    lab = fr["semantic"][f].reshape(-1).numpy()[ok].astype(np.int64)
    # map order (y flipped, x, z)
    k0 = M - 1 - k[:, 1]; r0 = 1 - r[:, 1]; k1 = k[:, 0]; r1 = r[:, 0]; k2 = k[:, 2]; r2 = r[:, 2]
    def foot(kk, rr):
        lo = np.where(rr < 0.5, np.maximum(kk - 1, 0), kk); hi = np.where(rr < 0.5, kk, np.minimum(kk + 1, M - 1)); return lo, hi
    a0, a1, a2 = foot(k0, r0), foot(k1, r1), foot(k2, r2)
    vox = np.stack([(x * M + y) * M + z for x in a0 for y in a1 for z in a2], 1)          # [P, 8]
    tile = np.stack([(((x >> S[0]) * nt[1] + (y >> S[1])) * nt[2] + (z >> S[2])) for x in a0 for y in a1 for z in a2], 1)
    vox_touched[vox.reshape(-1)] = True
    # records: distinct tiles per point
    ts = np.sort(tile, 1); first = np.ones_like(ts, bool); first[:, 1:] = ts[:, 1:] != ts[:, :-1]
    rec_cnt += np.bincount(ts[first], minlength=rec_cnt.size)
    con_cnt += np.bincount(tile.reshape(-1), minlength=rec_cnt.size)
    uv = np.unique(vox.reshape(-1))                       # cells of this frame = distinct voxels
    x = uv // (M * M); y = (uv // M) % M; z = uv % M
    cell_cnt += np.bincount(((x >> S[0]) * nt[1] + (y >> S[1])) * nt[2] + (z >> S[2]), minlength=rec_cnt.size)
    print("frame", f, "valid", int(ok.sum()), file=sys.stderr)
nz = rec_cnt > 0
rc, cc, ce = rec_cnt[nz], con_cnt[nz], cell_cnt[nz]
print("tiles", int(nz.sum()), "records", int(rc.sum()), "contribs", int(cc.sum()), "cells", int(ce.sum()), "union voxels", int(vox_touched.sum()))
for q in (10, 25, 50, 75, 83, 90, 95, 99, 100):
    print(f"p{q}: records {np.percentile(rc, q):.0f} contribs {np.percentile(cc, q):.0f} cells {np.percentile(ce, q):.0f}")
for thr in (64, 128, 192, 256, 384, 512, 1024, 2048):
    m = rc <= thr
    print(f"records<={thr}: tiles {m.mean():.3f} records {rc[m].sum() / rc.sum():.3f} contribs {cc[m].sum() / cc.sum():.3f} max cells {ce[m].max()} p99 cells {np.percentile(ce[m], 99):.0f} mean cells {ce[m].mean():.0f}")
np.savez("/tmp/distA_tile_stats.npz", rec=rec_cnt, con=con_cnt, cell=cell_cnt)

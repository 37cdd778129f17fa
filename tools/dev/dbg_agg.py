#!/usr/bin/env python3
"""dev: aggregated entries vs contributions vs oracle on the first case of tests/test_gpu_aggregated.py"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from test_gpu_cells import layers, run_both
from test_gpu_aggregated import room_frames
dev = torch.device("cuda:0")
H, W, M, n, C = 60, 80, 64, int(os.environ.get("N", 16)), 54
fr = room_frames(2 * n, H, W, C, seed=3)
out = {}
for fmt in ("aggregated", "contributions", "records"):
    os.environ["MF_FORMAT"] = fmt
    lay, ref = layers(dev, "label", C, H, W, M, 0.1, iw=0.5)
    for sl in (slice(0, n),):
        run_both(lay, ref, fr, sl, "label", C)
    out[fmt] = lay.data.cpu().numpy().astype(np.float64)
want = ref.data.numpy().astype(np.float64)
for fmt, got in out.items():
    err = np.abs(got - want)
    bad = err > 1e-4 * np.abs(want) + 1e-6
    print(fmt, "bad", int(bad.sum()), "max abs", err.max(), "occ equal", np.array_equal(got != 0, want != 0), "sum got", got.sum(), "want", want.sum())
    idx = np.argsort(err.ravel())[::-1][:8]
    for i in idx:
        p = np.unravel_index(i, got.shape)
        print("   ", p, "got", got[p], "want", want[p], "contrib-fmt", out["contributions"][p])

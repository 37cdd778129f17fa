"""dev: where does the cells kernel differ from the oracle on the >64-frame case?"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from test_gpu_cells import layers, sparse_frames, run_both
from conftest import last_fuse_mode
dev = torch.device("cuda:0")
import ast
for n, bad in ast.literal_eval(os.environ.get('CASES', '((150, False),)')):
    H, W, M, C = 30, 40, 48, 9
    lay, ref = layers(dev, "label", C, H, W, M, 0.12)
    fr = sparse_frames(n, H, W, C, seed=21, dmax=2.8)
    if bad:
        fr["semantic"][::7, ::5, ::3] = C
    run_both(lay, ref, fr, slice(0, n), "label", C)
    got, want = lay.data.cpu().numpy().astype(np.float64), ref.data.numpy().astype(np.float64)
    occ = (got != 0) != (want != 0)
    err = np.abs(got - want) > 1e-4 * np.abs(want) + 1e-6
    print("n", n, "bad ids", bad, "mode", last_fuse_mode(lay, n), "occupancy diffs", int(occ.sum()), "value diffs", int(err.sum()))
    idx = np.argwhere(occ)[:6]
    for i in idx:
        print("   ", tuple(i), "got", got[tuple(i)], "want", want[tuple(i)], "voxel got", got[tuple(i[:3])].round(6)[:9], "want", want[tuple(i[:3])].round(6)[:9])

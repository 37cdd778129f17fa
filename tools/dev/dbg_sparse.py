import sys, os, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import test_gpu_splat as T
splat = np.load(os.path.join(R, "tests/golden/splat_small.npz")); geom = np.load(os.path.join(R, "tests/golden/geom_small.npz"))
dev = torch.device("cuda:0")
for C, kind in ((1, "ones"), (5, "label")):
    lay = T.make_layer(C, kind, dev)
    lay.update_batch(T.batch_obs(splat, geom, C, kind), sequential=True)
    got = lay.data.cpu().numpy().astype(np.float64); want = splat[f"C{C}{kind}_seq2_map"].astype(np.float64)
    print(C, kind, "sum got", got.sum(), "want", want.sum(), "nnz got", (got != 0).sum(), "want", (want != 0).sum(),
          "maxerr", np.abs(got - want).max())
    bad = np.argwhere((got != 0) != (want != 0))
    print(" first bad", bad[:5].tolist(), " tiles of bad (y/4,x/4,z/8):", sorted(set((int(b[0]) // 4, int(b[1]) // 4, int(b[2]) // 8) for b in bad))[:12], "n bad tiles", len(set((int(b[0]) // 4, int(b[1]) // 4, int(b[2]) // 8) for b in bad)))

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
    err = np.abs(got - want); tol = 1e-4 * np.abs(want) + 1e-6
    print(C, kind, "sum got", got.sum(), "want", want.sum(), "nnz got", (got != 0).sum(), "want", (want != 0).sum(),
          "maxerr", err.max(), "viol", int((err > tol).sum()), "occ diff", int(((got != 0) != (want != 0)).sum()))

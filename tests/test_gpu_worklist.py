"""Work-list boundaries of the three tile kernels: the look-ups each workgroup makes ahead
of its current tile (ticket -> work list entry -> bucket offsets) must stay inside the listed tiles, the
scanned cursor array and the classes' sizes whatever the relation between the number of listed tiles L and
the number of workgroups B: L < B, L < 4 B (static dealing only), L a multiple of B, B = 1, and a list whose
tiles all sit in one or two load classes.  Each case runs in a fresh process (MF_BLOCKS and the kernel
overrides are read once per process) and is compared with the oracle."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

CASE = r"""
import sys, numpy as np, torch
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/tests")
from conftest import assert_map_close, last_fuse_mode
from test_gpu_cells import layers, sparse_frames, run_both
from mass_amd import _lib
dev = torch.device("cuda:0")
H, W, M, C, n = 48, 64, 32, 6, 10          # 32^3 map: 8 x 8 x 4 = 256 tiles of 4 x 4 x 8 (4 x 4 x 4 = 64 of 8 x 8 x 8)
for seed, dmax, spread in ((1, 1.2, 0.6), (2, 0.45, 0.02)):      # most tiles listed / a handful of tiles listed
    lay, ref = layers(dev, "label", C, H, W, M, 0.1)
    fr = sparse_frames(n, H, W, C, seed=seed, dmin=0.3, dmax=dmax, spread=spread)
    run_both(lay, ref, fr, slice(0, n), "label", C)
    mode = last_fuse_mode(lay, n)
    assert mode == %(mode)d, (mode, %(mode)d)
    assert_map_close(lay.data.cpu().numpy(), ref.data.numpy(), what="seed %%d" %% seed)
print("WORKLIST_OK")
"""


@pytest.mark.parametrize("kernel,env,mode", [("cells", {"MF_CELLS_FORCE": "1"}, 3), ("dense", {"MF_DENSE_FORCE": "1"}, 2),
                                             ("tiles", {"MF_DENSE": "0"}, 0)])
@pytest.mark.parametrize("blocks", [1, 64, 256, 300])
def test_work_list_boundaries(kernel, env, mode, blocks):
    e = dict(os.environ, PYTHONPATH=ROOT, MF_BLOCKS=str(blocks), **env)
    out = subprocess.run([sys.executable, "-c", CASE % dict(root=ROOT, mode=mode)], env=e, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0 and "WORKLIST_OK" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])

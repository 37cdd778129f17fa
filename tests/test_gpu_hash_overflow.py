"""The per-block bucket hash of the count / scatter kernels has a bounded probe length; entries
that do not fit fall back to one global atomic each.  Force that path (1 x 1 x 2-voxel tiles via
the MF_TILE tuning override, so a 256-pixel block sees > 1000 distinct buckets) in a fresh
process and compare with the oracle."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

SCRIPT = r"""
import numpy as np, torch, sys
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
from conftest import assert_map_close
from mass_amd.nn.base_projection_layer import BaseProjectionLayer
from oracle import massref as orc
kw = dict(camera_height=32, camera_width=48, map_height=40, map_width=40, map_depth=24, feature_size=3, grid_resolution=0.08)
lay = BaseProjectionLayer(**kw).cuda(); ol = orc.RefProjectionLayer(**kw)
g = torch.Generator().manual_seed(5)
n = 6
batch = dict(position=0.2 * torch.randn(n, 3, generator=g), yaw=6.28 * torch.rand(n, generator=g),
             elevation=-0.5 * torch.rand(n, generator=g), depth=0.3 + 1.5 * torch.rand(n, 32, 48, 1, generator=g),
             features=torch.rand(n, 32, 48, 3, generator=g))
lay.update_batch(batch, sequential=True)
for t in range(n):
    ol.update({k: v[t] for k, v in batch.items()})
assert_map_close(lay.data.cpu().numpy(), ol.data.numpy())
print("OVERFLOW-PATH-OK", int((ol.data != 0).sum()))
""" % (ROOT, ROOT)


def test_bucket_hash_overflow_path(device):
    env = dict(os.environ, MF_TILE="0 0 1 64")
    out = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "OVERFLOW-PATH-OK" in out.stdout

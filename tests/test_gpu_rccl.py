"""The run's one collective on the real backend: a SUM all-reduce of the metrics vector through
torch.distributed's "nccl" backend (= RCCL on ROCm) with a process group of world size 1 on cuda:0, in a
fresh process (the 8-GPU node is the driver's to launch; on this box RCCL has at least executed once),
and the 2-rank rehearsal of the episode workload (BASELINE configs[4]) sharing the one GPU over gloo."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

NCCL_PROG = r"""
import os, sys, json
sys.path.insert(0, %r)
import torch, torch.distributed as dist
os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MF_TEST_PORT", "29541"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from mass_amd import distributed as D
m = D.reduce_metrics(dict(frames=192.0, n_matches=3.0, map_abs_sum=1234.5))
t = torch.arange(16, dtype=torch.float64, device="cuda:0")
dist.all_reduce(t, op=dist.ReduceOp.SUM)                 # ncclAllReduce on one rank: the identity, but through RCCL
mx = D.max_over_ranks(2.5)
D.barrier()
print(json.dumps(dict(backend=dist.get_backend(), metrics=m, vec=t.tolist(), mx=mx, device=str(D._device()))))
dist.destroy_process_group()
"""


def test_rccl_all_reduce_with_a_one_rank_group():
    env = dict(os.environ, PYTHONPATH=ROOT)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, "-c", NCCL_PROG % ROOT], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["backend"] == "nccl" and rec["device"].startswith("cuda")
    assert rec["metrics"] == dict(frames=192.0, n_matches=3.0, map_abs_sum=1234.5)
    assert rec["vec"] == list(map(float, range(16))) and rec["mx"] == 2.5


def run_bench(extra, env_extra=None):
    env = dict(os.environ, PYTHONPATH=ROOT, **(env_extra or {}))
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "episode", "--episodes", "2",
                          "--episode-frames", "12", "--batch", "12", "--steps", "1", "--warmup", "0"] + extra,
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_episode_workload_two_ranks_equal_one_rank():
    """bench.py --workload episode: 2 episodes (walkthrough + unshuffle map each, full 480x640 -> 256^3 x 54, a short
    trajectory) on one rank, and sharded over two ranks that share this box's GPU (gloo rehearsal): the all-reduced
    counters agree (the integer kernels give identical maps, so even the map sums are equal), and the moved object
    is found in both episodes."""
    one = run_bench(["--gpus", "1"])
    two = run_bench(["--gpus", "2"], {"MF_BENCH_BACKEND": "gloo"})
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    m1, m2 = one["metrics_allreduce"], two["metrics_allreduce"]
    assert m1["episodes"] == m2["episodes"] == 2 and m1["frames"] == m2["frames"] == 2 * 2 * 12
    for k in ("moved_found", "n_matches", "occupied_voxels"):
        assert m1[k] == m2[k], k
    np.testing.assert_allclose(m2["map_abs_sum"], m1["map_abs_sum"], rtol=1e-9)
    np.testing.assert_allclose(m2["shift_m"], m1["shift_m"], rtol=1e-5)
    assert m1["moved_found"] == 2, "the moved object class is found in every episode"
    assert one["scaling"] == "strong" and one["value"] > 0

"""Multi-GPU glue: one process per GPU, episodes sharded across ranks, and the
single collective of a run — a SUM all-reduce of a small fp64 metrics vector.

The reference shards the same way with --start-task/--every-tasks over separate
OS processes and merges per-task JSON files offline
(/root/reference/agent.py:154-155,661-662; create_submission.py:25-42); here the
merge is one RCCL all-reduce (`nccl` backend on ROCm), or `gloo` on CPU in tests.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Join the job described by RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns
    (rank, world_size, local_rank).  A single process needs no process group."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kwargs = {}
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            kwargs["device_id"] = torch.device("cuda", local_rank)
        dist.init_process_group(backend, **kwargs)
    return rank, world, local_rank


def shard(items, rank, world_size):
    """Item i runs on rank i mod world_size."""
    return [x for i, x in enumerate(items) if i % world_size == rank]


def _device():
    if dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def reduce_metrics(metrics):
    """Sum a {name: number} dict over all ranks (fp64); every rank gets the totals."""
    names = sorted(metrics)
    vec = torch.tensor([float(metrics[k]) for k in names], dtype=torch.float64, device=_device())
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(vec, op=dist.ReduceOp.SUM)
    return dict(zip(names, vec.tolist()))


def max_over_ranks(value):
    t = torch.tensor([float(value)], dtype=torch.float64, device=_device())
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if torch.cuda.is_available() and dist.is_initialized() and dist.get_backend() == "nccl":
        torch.cuda.synchronize()
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
    if torch.cuda.is_available() and dist.is_initialized() and dist.get_backend() == "nccl":
        torch.cuda.synchronize()

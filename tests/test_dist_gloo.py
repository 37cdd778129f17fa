"""The N > 1 path on CPU: two gloo ranks shard episodes and all-reduce their
metrics; the totals must equal the single-process run."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp


def episode_metrics(e):
    """Stand-in for one episode's counters (the real ones come from the GPU path):
    deterministic in the episode id."""
    g = torch.Generator().manual_seed(1000 * e)
    return dict(frames=300.0, valid_points=float(torch.randint(1, 10 ** 6, (1,), generator=g)),
                touched_voxels=float(torch.randint(1, 10 ** 6, (1,), generator=g)),
                map_abs_sum=float(torch.rand((), generator=g, dtype=torch.float64)) * 1e5)


def total(episodes):
    out = {}
    for e in episodes:
        for k, v in episode_metrics(e).items():
            out[k] = out.get(k, 0.0) + v
    return out


def worker(rank, world, port, n_episodes, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from mass_amd import distributed as D
    from mass_amd.episodes import shard_episodes
    r, w, _ = D.init_from_env(backend="gloo")
    mine = shard_episodes(n_episodes, r, w)
    assert mine == D.shard(list(range(n_episodes)), r, w)
    local = total(mine) if mine else {k: 0.0 for k in episode_metrics(0)}
    D.barrier()
    summed = D.reduce_metrics(local)
    slowest = D.max_over_ranks(1.0 + rank)
    q.put((rank, mine, summed, slowest))
    D.barrier()
    torch.distributed.destroy_process_group()


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_rank_episode_sharding_and_metrics_allreduce():
    world, n_episodes = 2, 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, n_episodes, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = total(range(n_episodes))
    seen = []
    for rank, mine, summed, slowest in results:
        seen += mine
        assert slowest == float(world)
        for k in want:
            np.testing.assert_allclose(summed[k], want[k], rtol=1e-12)
    assert sorted(seen) == list(range(n_episodes))       # every episode on exactly one rank


def test_single_process_needs_no_group():
    from mass_amd import distributed as D
    assert D.reduce_metrics(dict(a=1.5, b=2)) == dict(a=1.5, b=2.0)
    assert D.max_over_ranks(3) == 3.0
    assert D.shard([10, 11, 12, 13, 14], 1, 2) == [11, 13]

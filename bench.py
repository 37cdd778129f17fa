#!/usr/bin/env python3
"""bench.py — RGB-D frames/s fused into a 256^3 x 54-class semantic voxel map.

    python bench.py --gpus N --steps K --warmup W        (N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step is one pass of the hot path (mf_fuse_frames: unproject + bin + tile
scatter + blend, sequential semantics = 64 successive layer.update() calls)
over one batch of 64 synthetic 480x640 depth + uint8-label frames, inputs
resident in HBM (BASELINE.json configs[1], distribution A of SURVEY 8(d)).
Each rank owns its own map and its own frames (independent episodes, weak
scaling); the only collective is the final metrics all-reduce.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
H, W, C, MAP, BATCH = 480, 640, 54, 256, 64


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--workload", default="distA", choices=["distA", "room"])
    ap.add_argument("--cpu-frames", type=int, default=12, help="frames of the batch timed through the CPU oracle")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def make_frames(workload, n, seed0):
    from mass_amd.episodes import dist_a_frames, room_trajectory
    if workload == "distA":
        return dist_a_frames(n, seed0=seed0, height=H, width=W)
    tr = room_trajectory(n, H, W, seed=seed0)
    return {k: tr[k] for k in ("position", "yaw", "elevation", "depth", "semantic")}


def touched_per_frame(lay, poses, depth):
    """T_f = distinct voxels in the 8-corner footprint of frame f (the T of the
    algorithmic-bytes formula), from the HIP integer outputs (tests prove them
    bit-identical to the oracle's).  Not timed."""
    from mass_amd.utils.projection import unproject_bin
    s = (lay.map_height, lay.map_width, lay.map_depth)
    T, valid_pts = [], 0
    for f in range(depth.shape[0]):
        ix, iy, iz, rx, ry, rz, valid = unproject_bin(lay.bins_x, lay.bins_y, lay.bins_z, lay.rays,
                                                      poses[f:f + 1], depth[f:f + 1])
        v = valid.bool()
        valid_pts += int(v.sum())
        axes = []
        for k, r, size in ((iy[v], ry[v], s[0]), (ix[v], rx[v], s[1]), (iz[v], rz[v], s[2])):
            lo = torch.where(r < 0.5, (k - 1).clamp(min=0), k)
            hi = torch.where(r < 0.5, k, (k + 1).clamp(max=size - 1))
            axes.append((lo, hi))
        ids = torch.cat([((a * s[1] + b) * s[2] + c) for a in axes[0] for b in axes[1] for c in axes[2]])
        T.append(int(torch.unique(ids).numel()))
    return T, valid_pts


def cpu_baseline(frames, n):
    """The oracle (oracle/massref.c, scalar C port of the reference algorithm,
    1 thread) on the first n frames of rank 0's batch, sequential, same map size."""
    from oracle import massref as orc
    lay = orc.RefProjectionLayer(camera_height=H, camera_width=W, map_height=MAP, map_width=MAP, map_depth=MAP,
                                 feature_size=C, grid_resolution=0.05)
    obs = []
    for f in range(n):
        obs.append(dict(position=frames["position"][f], yaw=frames["yaw"][f], elevation=frames["elevation"][f],
                        depth=frames["depth"][f],
                        features=torch.nn.functional.one_hot(frames["semantic"][f].long(), C).float()))
    t0 = time.perf_counter()
    for o in obs:
        lay.update(o)
    dt = time.perf_counter() - t0
    return dict(value=n / dt, unit="frames/s", cores=1, kind="port",
                sample=f"first {n} frames of rank 0's batch through oracle/massref.c (bin_rays + "
                       f"update_feature_map, one-hot fp32 features as the reference builds them), "
                       f"sequential, {MAP}^3 x {C}, {dt:.1f} s"), lay


def main():
    args = parse()
    from mass_amd import distributed as D
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env == 1 and args.gpus > 1:
        raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
    # nccl == RCCL on ROCm.  MF_BENCH_BACKEND=gloo is a rehearsal knob for boxes with fewer GPUs
    # than ranks (ranks then share device rank % device_count); the driver never sets it.
    backend = os.environ.get("MF_BENCH_BACKEND", "nccl")
    n_dev = max(torch.cuda.device_count(), 1)
    if backend == "nccl":
        rank, world, local_rank = D.init_from_env(backend="nccl")
    else:
        rank, world, local_rank = D.init_from_env(backend=backend)
        local_rank %= n_dev
    args.gpus = world
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from mass_amd import _lib
    from mass_amd.nn.applications.semantic_projection_layer import SemanticProjectionLayer

    lay = SemanticProjectionLayer(camera_height=H, camera_width=W, map_height=MAP, map_width=MAP, map_depth=MAP,
                                  feature_size=C, grid_resolution=0.05).train().to(dev)
    frames = make_frames(args.workload, args.batch, seed0=rank * args.batch)
    poses = lay._poses(frames["position"], frames["yaw"], frames["elevation"])
    depth = frames["depth"].to(dev).reshape(args.batch, H, W).contiguous()
    label = frames["semantic"].to(dev).contiguous()

    from mass_amd.utils.projection import fuse_frames

    def step():
        fuse_frames(lay.bins_x, lay.bins_y, lay.bins_z, lay.rays, poses, depth, label, lay.data,
                    interpolation_weight=lay.interpolation_weight, sequential=True, workspace=lay._workspace)

    def barrier():
        torch.cuda.synchronize()
        D.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    _lib.check(_lib.lib.mf_profile_enable(1))
    n_prof = min(args.steps, 256)            # the library keeps events for 256 calls
    stage_ms = np.zeros((n_prof, 5), np.float32)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for k in range(args.steps):
        step()                     # stage events are recorded on the stream, nothing synchronises here
    ev1.record()
    barrier()
    wall = time.perf_counter() - t0
    for k in range(n_prof):
        _lib.check(_lib.lib.mf_profile_read(k, stage_ms[k].ctypes.data))
    _lib.check(_lib.lib.mf_profile_enable(0))
    gpu_ms = ev0.elapsed_time(ev1)

    wall_max = D.max_over_ranks(wall)

    # ---- algorithmic bytes of one step: sum_f H*W*(4+1) + T_f*C*4*2 ------------
    T, valid_pts = touched_per_frame(lay, poses, depth)
    alg_bytes = sum(H * W * (4 + 1) + Tf * C * 4 * 2 for Tf in T)
    # the one data collective of the run: SUM all-reduce of the per-rank counters (RCCL)
    metrics = D.reduce_metrics(dict(frames=args.batch * args.steps, valid_points=valid_pts * args.steps,
                                    touched_voxels=sum(T) * args.steps,
                                    map_abs_sum=float(lay.data.abs().sum(dtype=torch.float64))))

    if rank == 0:
        frames_total = args.batch * args.steps * world
        ms_per_step = wall_max / args.steps * 1e3
        fuse_ms = float(stage_ms[:, 3].mean())
        step_ms = float(stage_ms[:, 4].mean())
        achieved = alg_bytes / (fuse_ms * 1e-3) / 1e9
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tfile):
            with open(tfile) as f:
                traffic = json.load(f).get(f"{args.workload}_b{args.batch}")
        out = {
            "metric": "RGB-D frames/s fused into 256^3 semantic voxel map",
            "value": frames_total / wall_max, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"configs[1]: {args.batch} x 480x640 depth + u8 54-class labels -> 256^3 x 54 "
                                   f"fp32 map at 0.05 m, distribution {'A (depth 0.5+4.5U, random poses)' if args.workload == 'distA' else 'B (box room trajectory)'}, "
                                   f"sequential blend (= {args.batch} layer.update calls), one map per GPU",
                       "frames_per_step": args.batch, "map": [MAP, MAP, MAP, C], "mode": "sequential"},
            "roofline": {"bound": "hbm", "kernel": "fuse_tiles_kernel<1>", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": fuse_ms,
                         "traffic_GBps": (traffic / (fuse_ms * 1e-3) / 1e9) if traffic else None,
                         "traffic_frac_of_peak": (traffic / (fuse_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                         "note": "algorithmic bytes = sum over the launch's frames of H*W*5 + T_f*C*8 (per-frame "
                                 "RMW of every touched voxel); the kernel keeps map tiles in LDS across the 64 "
                                 "frames, so HBM sees each touched voxel once per launch: `traffic` (rocprofv3 PMC, "
                                 "profiles/) is the bytes that actually moved, and frac > 1 means the launch beats what "
                                 "a per-frame implementation could do at 100% of HBM peak"},
            "roofline_step": {"achieved": alg_bytes / (step_ms * 1e-3) / 1e9, "frac": alg_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "step_ms_gpu": step_ms, "stage_ms": {"zero+count": float(stage_ms[:, 0].mean()),
                                                                   "scan": float(stage_ms[:, 1].mean()),
                                                                   "scatter": float(stage_ms[:, 2].mean()),
                                                                   "fuse_tiles": fuse_ms}},
            "touched_voxels_per_frame_mean": float(np.mean(T)),
            "gpu_ms_total": gpu_ms,
            "metrics_allreduce": metrics,
        }
        if world == 1 and not args.no_cpu_baseline:
            cb, _ = cpu_baseline(frames, args.cpu_frames)
            cb["host_cpus"] = os.cpu_count()
            out["cpu_baseline"] = cb
        print(json.dumps(out), flush=True)
    if world > 1:
        D.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

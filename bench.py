#!/usr/bin/env python3
"""bench.py — RGB-D frames/s fused into a 256^3 x 54-class semantic voxel map.

    python bench.py --gpus N --steps K --warmup W

N = 1 runs in this process.  N > 1 without WORLD_SIZE in the environment starts
N fresh rank processes (python -m torch.distributed.run) BEFORE this process has
made any GPU call and relays rank 0's JSON line; under torch.distributed.run it
is one rank of the job (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the env).

A step is one pass of the hot path (mf_fuse_frames: unproject + bin + tile
scatter + blend, sequential semantics = 64 successive layer.update() calls)
over one batch of 64 synthetic 480x640 depth + uint8-label frames, inputs
resident in HBM (BASELINE.json configs[1], distribution A of SURVEY 8(d)).
Each rank owns its own map and its own frames (independent episodes, weak
scaling); the only collective is the final metrics all-reduce.

Roofline accounting: a launch keeps every map tile in LDS
across its 64 frames, so the bytes that MUST move per launch are
    inputs  B*H*W*(4 + 1)    +    |union of the frames' 8-corner footprints| * C * 4 * 2
(each touched voxel read once and written once).  `roofline.frac` prices the
tile kernel against exactly that (never above 1); SURVEY 8(d)'s per-frame figure
(a read-modify-write per frame) is reported as `per_frame_model`, informational.

Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import socket
import statistics
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
H, W, C, MAP, BATCH = 480, 640, 54, 256, 64
KERNEL_SOURCES = ("mass_amd/csrc/fuse.hip", "mass_amd/csrc/geometry.h")


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--workload", default="distA", choices=["distA", "room", "episode"],
                    help="distA / room: configs[1] batches (distribution A is the headline); episode: configs[4], every rank "
                         "runs its share of --episodes synthetic episodes (two 300-frame maps + matching each), a step = one round")
    ap.add_argument("--episodes", type=int, default=8)
    ap.add_argument("--episode-frames", type=int, default=300)
    ap.add_argument("--rotate", type=int, default=4,
                    help="distinct resident batches per rank, taken in turn (inputs + records exceed the 256 MB Infinity Cache)")
    ap.add_argument("--mode", default="sequential", choices=["sequential", "merged"])
    ap.add_argument("--cpu-frames", type=int, default=6, help="frames per repetition of the CPU oracle (1 warm + 3 timed)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the other SURVEY 8(d) workloads (rank 0, N = 1)")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="issue every step as one mf_fuse_frames call instead of overlapping the bucketing of step k+1 "
                         "with the tile kernels of step k (mf_fuse_frames_stage / _commit on two streams)")
    ap.add_argument("--rank-seed-stride", type=int, default=-1,
                    help="rank r draws its frames from seeds r*stride.. (default: the batch size, i.e. disjoint "
                         "episodes per rank; 0 gives every rank the same frames, used by the 2-rank rehearsal test)")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------
# N > 1 from a plain `python bench.py --gpus N`: start the ranks as children (no GPU call here)
# ----------------------------------------------------------------------------------------------
def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args, argv):
    """The reference shards episodes over separate OS processes (agent.py:154-155,795-800);
    so does this: N children, one per GPU, started before the parent touches the device."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    proc = subprocess.run(cmd, env=env)
    return proc.returncode


# ----------------------------------------------------------------------------------------------
# workload helpers
# ----------------------------------------------------------------------------------------------
def make_frames(workload, n, seed0):
    from mass_amd.episodes import dist_a_frames, room_trajectory
    if workload == "distA":
        return dist_a_frames(n, seed0=seed0, height=H, width=W)
    tr = room_trajectory(n, H, W, seed=seed0)
    return {k: tr[k] for k in ("position", "yaw", "elevation", "depth", "semantic")}


def footprints(lay, poses, depth):
    """Per frame T_f = distinct voxels in the frame's 8-corner footprint, and the size of the
    UNION of those sets over the batch (the voxels one fused launch has to read and write),
    from the HIP integer outputs (tests prove them bit-identical to the oracle's).  Not timed."""
    from mass_amd.utils.projection import unproject_bin
    s = (lay.map_height, lay.map_width, lay.map_depth)
    seen = torch.zeros(s[0] * s[1] * s[2], dtype=torch.bool, device=depth.device)
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
        ids = torch.unique(torch.cat([((a * s[1] + b) * s[2] + c) for a in axes[0] for b in axes[1] for c in axes[2]]))
        T.append(int(ids.numel()))
        seen[ids] = True
    return T, valid_pts, int(seen.sum())


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(frames, per_rep, reps=3):
    """The oracle (oracle/massref.c, C port of the reference algorithm) on the first (1 + reps) * per_rep frames
    of rank 0's batch, sequential, same map size, on all host threads it may use (its blend loop is threaded so that
    the result is bit-identical to the single thread): one warm repetition, then `reps` timed ones; the median rate
    is `value`.  The single-thread rate of the same code is measured on a few more frames and reported beside it.
    Returns the baseline record, the oracle layer (its map after the multi-thread frames is what the GPU parity
    check compares with) and the number of frames it holds."""
    from oracle import massref as orc
    lay = orc.RefProjectionLayer(camera_height=H, camera_width=W, map_height=MAP, map_width=MAP, map_depth=MAP,
                                 feature_size=C, grid_resolution=0.05)
    threads = orc.get_threads()
    n = (1 + reps) * per_rep

    def obs(f):
        return dict(position=frames["position"][f], yaw=frames["yaw"][f], elevation=frames["elevation"][f],
                    depth=frames["depth"][f], features=torch.nn.functional.one_hot(frames["semantic"][f].long(), C).float())
    rates, total = [], 0.0
    for r in range(1 + reps):
        batch = [obs(f) for f in range(r * per_rep, (r + 1) * per_rep)]        # the one-hot images are built outside the timing
        t0 = time.perf_counter()
        for o in batch:
            lay.update(o)
        dt = time.perf_counter() - t0
        total += dt
        if r > 0:
            rates.append(per_rep / dt)
    used = orc.last_threads()          # 1 on unrelated frames: memory bound, the port takes one thread there (see massref.c)
    # the other thread count, on a copy of the map (so that the parity map stays the one after n frames)
    other, other_threads = None, (threads if used == 1 else 1)
    n1 = min(3, max(frames["depth"].shape[0] - n, 0))
    if n1 > 0 and other_threads != used:
        lay1 = orc.RefProjectionLayer(camera_height=H, camera_width=W, map_height=MAP, map_width=MAP, map_depth=MAP,
                                      feature_size=C, grid_resolution=0.05)
        lay1.data.copy_(lay.data)
        batch = [obs(f) for f in range(n, n + n1)]
        orc.force_threads(True)
        orc.set_threads(other_threads)
        try:
            t0 = time.perf_counter()
            for o in batch:
                lay1.update(o)
            other = n1 / (time.perf_counter() - t0)
        finally:
            orc.force_threads(False)
            orc.set_threads(threads)
        del lay1
    rec = dict(value=statistics.median(rates), unit="frames/s", cores=used, kind="port",
               reps=reps, rates=[round(x, 4) for x in rates],
               other_thread_count=dict(threads=other_threads, frames_per_s=other),
               cpu_model=cpu_model(), host_cpus=os.cpu_count(),
               torch_threads=torch.get_num_threads(), torch=torch.__version__,
               sample=f"frames {per_rep}..{n - 1} of rank 0's batch through oracle/massref.c (bin_rays + "
                      f"update_feature_map on one-hot fp32 features as the reference builds them), sequential "
                      f"onto one {MAP}^3 x {C} map: 1 warm + {reps} timed repetitions of {per_rep} frames, median; "
                      f"{total:.1f} s of CPU work on {used} thread(s) (the port threads its blend loop over the touched voxels, "
                      f"bit-identically, when voxels receive many contributions each; on unrelated frames that loop is "
                      f"memory bound and it takes one thread); other_thread_count: the next {n1} frames forced onto "
                      f"{other_threads} thread(s)",
               reference_in_build_container="0.52 frames/s (the reference's own torch CPU path, 8 threads, "
                                            "256^3 x 54; SURVEY section 6)")
    return rec, lay, n


MODE_NAMES = {0: "fuse_tiles_kernel", 2: "fuse_dense_kernel", 3: "fuse_cells_kernel", 4: "fuse_cells_kernel<aggregated entries>"}


def last_mode(lay, n_frames, ws=None):
    """Which tile kernel took the last sequential multi-frame call on this layer's (or the given) workspace."""
    from mass_amd import _lib
    from mass_amd.utils.projection import _grid_struct
    g = _grid_struct(lay.data, lay.bins_x, lay.bins_y, lay.bins_z)
    wptr, _ = (ws or lay._workspace).get(1, lay.data.device)
    m = _lib.lib.mf_fuse_last_mode(g, n_frames * H * W, n_frames, wptr, _lib.current_stream(lay.data.device))
    return MODE_NAMES.get(m, str(m))


def parity_vs_oracle(lay_kw, frames, n, ref_layer, dev, pipelined):
    """Untimed: the first n frames of the batch onto a fresh map through the SAME issue path the timed steps take
    (pipelined: two FusePipeline submits of n / 2 frames, stage on the side stream + commit alone; else one
    mf_fuse_frames call), compared with the oracle map of the same frames."""
    from mass_amd.nn.applications.semantic_projection_layer import SemanticProjectionLayer
    from mass_amd.utils.projection import FusePipeline
    lay = SemanticProjectionLayer(**lay_kw).to(dev)
    depth, sem = frames["depth"][:n].to(dev).reshape(n, H, W), frames["semantic"][:n].to(dev)
    if pipelined and n >= 4:
        pipe, h = FusePipeline(dev), n // 2
        for sl in (slice(0, h), slice(h, n)):
            poses = lay._poses(frames["position"][sl], frames["yaw"][sl], frames["elevation"][sl])
            pipe.submit(lay.bins_x, lay.bins_y, lay.bins_z, lay.rays, poses, depth[sl], sem[sl], lay.data,
                        interpolation_weight=lay.interpolation_weight, sequential=True)
        pipe.flush()
        kernel = last_mode(lay, n - h, pipe.ws[1])
    else:
        lay.update_batch(dict(position=frames["position"][:n], yaw=frames["yaw"][:n], elevation=frames["elevation"][:n],
                              depth=depth, semantic=sem), sequential=True)
        kernel = last_mode(lay, n)
    torch.cuda.synchronize()
    occ_equal, ok, rel, occupied = True, True, 0.0, 0
    for y0 in range(0, MAP, 32):                       # compared on the device, slab by slab, in fp64
        got = lay.data[y0:y0 + 32].to(torch.float64)
        want = ref_layer.data[y0:y0 + 32].to(dev).to(torch.float64)
        occ_equal &= bool(torch.equal(got != 0, want != 0))
        err = (got - want).abs()
        ok &= bool((err <= 1e-4 * want.abs() + 1e-6).all())
        rel = max(rel, float((err / (want.abs() + 1e-6)).max()))
        occupied += int((want != 0).any(-1).sum())
    del lay
    return dict(parity_checked_frames=n, occupancy_bit_exact=occ_equal, within_tolerance=ok,
                tolerance="|got - want| <= 1e-4 |want| + 1e-6", max_scaled_err=rel, occupied_voxels=occupied,
                tile_kernel=kernel, issue="pipelined (two stage / commit pairs)" if pipelined and n >= 4 else "one call",
                note="GPU map after the first n frames, issued like the timed steps, vs the oracle map of the same frames")


def oracle_map(frames, n):
    """The oracle's map after the first n frames (threads as available: bit-identical to one thread)."""
    from oracle import massref as orc
    lay = orc.RefProjectionLayer(camera_height=H, camera_width=W, map_height=MAP, map_width=MAP, map_depth=MAP,
                                 feature_size=C, grid_resolution=0.05)
    for f in range(n):
        lay.update(dict(position=frames["position"][f], yaw=frames["yaw"][f], elevation=frames["elevation"][f],
                        depth=frames["depth"][f], features=torch.nn.functional.one_hot(frames["semantic"][f].long(), C).float()))
    return lay


def copy_bandwidth(dev, nbytes=1 << 30, reps=5):
    """Achieved HBM copy bandwidth of this GPU in this run (GB/s, read + write bytes): a 1 GiB device-to-device copy."""
    a = torch.empty(nbytes // 4, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)
    b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    del a, b
    return 2.0 * nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def sources_sha():
    h = hashlib.sha256()
    for p in KERNEL_SOURCES:
        with open(os.path.join(ROOT, p), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def recorded_traffic(key):
    """HBM bytes per launch from the rocprofv3 PMC passes (profiles/traffic.json), only if they
    were measured on the kernel sources this run uses."""
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tfile):
        return None, "profiles/traffic.json missing"
    with open(tfile) as f:
        t = json.load(f)
    if t.get("kernel_sources_sha16") != sources_sha():
        return None, (f"profiles/traffic.json was measured on kernel sources {t.get('kernel_sources_sha16')}, "
                      f"this run uses {sources_sha()}: not reported")
    return t.get(key), t.get(key + "_source", t.get("_source"))


def timed_fuse(lay, poses, depth, label, sequential, steps, warmup):
    """steps calls of the fused pipeline on resident inputs; returns (wall s, stage ms [steps, 5])."""
    from mass_amd import _lib
    from mass_amd.utils.projection import fuse_frames

    def step():
        fuse_frames(lay.bins_x, lay.bins_y, lay.bins_z, lay.rays, poses, depth, label, lay.data,
                    interpolation_weight=lay.interpolation_weight, sequential=sequential, workspace=lay._workspace)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    _lib.check(_lib.lib.mf_profile_enable(1))
    n_prof = min(steps, 256)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ms = np.zeros((n_prof, 5), np.float32)
    for k in range(n_prof):
        _lib.check(_lib.lib.mf_profile_read(k, ms[k].ctypes.data))
    _lib.check(_lib.lib.mf_profile_enable(0))
    return wall, ms


def stage_dict(ms):
    m = ms.mean(0)
    return {"zero+count": float(m[0]), "scan": float(m[1]), "scatter": float(m[2]), "fuse_tiles": float(m[3]),
            "call": float(m[4])}


def extra_workloads(lay_kw, dev, frames_a, poses_a, depth_a, label_a):
    """The other workloads SURVEY 8(d) specifies, each measured here on the same binary
    (rank 0, N = 1, after the headline; none of them enters `value`)."""
    from mass_amd.episodes import room_trajectory
    from mass_amd.nn.base_projection_layer import BaseProjectionLayer
    from mass_amd.nn.applications.occupancy_projection_layer import OccupancyProjectionLayer
    from mass_amd.nn.applications.semantic_projection_layer import SemanticProjectionLayer
    from mass_amd.utils.experimentation import pairwise_distance, linear_sum_assignment
    out = {}

    def batch_run(name, frames, poses, depth, label, sequential, steps=20):
        lay = SemanticProjectionLayer(**lay_kw).to(dev)
        wall, ms = timed_fuse(lay, poses, depth, label, sequential, steps, 3)
        T, valid, union = footprints(lay, poses, depth)
        B = depth.shape[0]
        step_bytes = B * H * W * 5 + union * C * 8
        st = stage_dict(ms)
        # the same steps issued like the headline: bucketing of step k+1 beside the tile kernel of step k
        from mass_amd.utils.projection import FusePipeline
        pipe = FusePipeline(dev)

        def pstep():
            pipe.submit(lay.bins_x, lay.bins_y, lay.bins_z, lay.rays, poses, depth, label, lay.data,
                        interpolation_weight=lay.interpolation_weight, sequential=sequential)
        for _ in range(4):
            pstep()
        pipe.flush()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            pstep()
        pipe.flush()
        torch.cuda.synchronize()
        wall_p = time.perf_counter() - t0
        del pipe
        out[name] = dict(tile_kernel=last_mode(lay, B if sequential else 1) if sequential else "fuse_dense_kernel / single-pass (merged: one group)",
                         frames_per_s=B * steps / wall, ms_per_step=wall / steps * 1e3, stage_ms=st,
                         frames_per_s_pipelined=B * steps / wall_p, ms_per_step_pipelined=wall_p / steps * 1e3,
                         union_voxels=union, touched_voxels_per_frame_mean=float(np.mean(T)),
                         algorithmic_bytes_per_launch=step_bytes,
                         frac_of_hbm_peak_kernel=union * C * 8 / (st["fuse_tiles"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         frac_of_hbm_peak_step=step_bytes / (st["call"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         mode="sequential" if sequential else "merged")
        del lay

    # configs[1], distribution A, merged mode (the functional API's batch semantics)
    batch_run("distA_merged_b64", frames_a, poses_a, depth_a, label_a, False)

    # configs[1], distribution B: 64 consecutive frames of the box-room trajectory, sequential
    tr = room_trajectory(BATCH, H, W, seed=0)
    probe = SemanticProjectionLayer(**lay_kw).to(dev)
    poses_b = probe._poses(tr["position"], tr["yaw"], tr["elevation"])
    del probe
    depth_b = tr["depth"].to(dev).reshape(BATCH, H, W).contiguous()
    label_b = tr["semantic"].to(dev).contiguous()
    batch_run("distB_sequential_b64", tr, poses_b, depth_b, label_b, True)
    del depth_b, label_b
    # ... and its parity at this shape, through the pipelined issue path: first 16 frames vs the oracle
    ref_b = oracle_map(tr, 16)
    out["distB_sequential_b64"]["parity"] = parity_vs_oracle(lay_kw, tr, 16, ref_b, dev, True)
    del ref_b

    # configs[2]: 300-frame room trajectory, three maps updated per frame through layer.update()
    # (occupancy C = 1, semantic C = 54, "RGB" C = 3 dense), as agent.py:107-111 drives them
    n3 = 300
    tr = room_trajectory(n3, H, W, seed=1)
    occ = OccupancyProjectionLayer(**{k: v for k, v in lay_kw.items() if k != "feature_size"}).to(dev)
    sem = SemanticProjectionLayer(**lay_kw).to(dev)
    rgb = BaseProjectionLayer(**dict(lay_kw, feature_size=3)).to(dev)
    d_dev, s_dev, c_dev = tr["depth"].to(dev), tr["semantic"].to(dev)[..., None], tr["rgb"].to(dev)

    from mass_amd.nn.feature_maps import update_feature_maps
    maps = dict(occupancy=occ, semantic=sem, rgb=rgb)
    d_np, s_np, c_np = tr["depth"].numpy(), tr["semantic"].numpy()[..., None].astype(np.int64), tr["rgb"].numpy()
    # the observation of every step, as the simulator wrapper hands it over (built outside the timed loops)
    obs_dev = [dict(position=tr["position"][t], yaw=tr["yaw"][t], elevation=tr["elevation"][t], depth=d_dev[t],
                    semantic=s_dev[t], features=c_dev[t]) for t in range(n3)]
    obs_host = [dict(position=tr["position"][t].numpy(), yaw=float(tr["yaw"][t]), elevation=float(tr["elevation"][t]),
                     depth=d_np[t], semantic=s_np[t], features=c_np[t]) for t in range(n3)]

    def run_traj(host_fed, validate=True, shared=True):
        for lay in (occ, sem, rgb):
            lay.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for o in (obs_host if host_fed else obs_dev):
            if shared:        # the agent's loop over its maps as one call (navigation_policy.py:164-171)
                update_feature_maps(maps, o, validate=validate)
            else:
                occ.update(o)
                sem.update(o, validate=validate)
                rgb.update(o)
        torch.cuda.synchronize()
        sem.check_labels()
        return time.perf_counter() - t0
    run_traj(False, validate="defer")
    dt = min(run_traj(False, validate="defer") for _ in range(2))
    dt_sync = min(run_traj(False, validate=True) for _ in range(2))
    run_traj(False, validate="defer", shared=False)
    dt_loop = min(run_traj(False, validate="defer", shared=False) for _ in range(2))
    dt_host = min(run_traj(True, validate="defer") for _ in range(2))
    dt_host_loop = run_traj(True, validate="defer", shared=False)
    out["config3_trajectory_300x3maps"] = dict(
        frames_per_s=n3 / dt, ms_per_frame_3_maps=dt / n3 * 1e3, updates_per_s=3 * n3 / dt,
        frames_per_s_synchronous_label_check=n3 / dt_sync,
        frames_per_s_layer_loop=n3 / dt_loop, ms_per_frame_layer_loop=dt_loop / n3 * 1e3,
        host_fed_frames_per_s=n3 / dt_host, host_fed_ms_per_frame=dt_host / n3 * 1e3,
        host_fed_frames_per_s_layer_loop=n3 / dt_host_loop,
        note="per simulator step the occupancy (C=1), semantic (C=54 labels) and RGB (C=3 dense fp32) maps, 256^3 each, "
             "take the same observation: frames_per_s through mass_amd.nn.update_feature_maps (one mf_fuse_frame_maps "
             "call: the frame is bucketed once, the maps' tile kernels run side by side), *_layer_loop through three "
             "layer.update() calls like the reference's loop; observations resident in HBM, class-id check deferred "
             "(validate='defer': an id out of range calls the map's update off on the device and raises at the next "
             "call into the layer); frames_per_s_synchronous_label_check with validate=True (one stream wait per step, "
             "raises from the call itself like the reference's one_hot); host_fed_* with numpy observations uploaded "
             "per step (PCIe inclusive, int64 label image as the simulator produces it)")
    del occ, sem, rgb, d_dev, s_dev, c_dev

    # configs[3]: matching, 200 x 200 instance pairs, 1024-d (experimentation.py:261-287)
    g = torch.Generator().manual_seed(0)
    f0 = torch.randn(200, 1024, generator=g).to(dev)
    f1 = torch.randn(200, 1024, generator=torch.Generator().manual_seed(1)).to(dev)
    res = {}
    for metric in ("l2", "l2_gemm"):
        for _ in range(3):
            pairwise_distance(f0, f1, metric)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            cost = pairwise_distance(f0, f1, metric)
        e1.record()
        torch.cuda.synchronize()
        res[f"pairwise_{metric}_ms"] = e0.elapsed_time(e1) / 50
    cost_h = cost.cpu().numpy().astype(np.float64)
    t0 = time.perf_counter()
    for _ in range(20):
        rows, cols = linear_sum_assignment(cost_h)
    res["assignment_host_ms"] = (time.perf_counter() - t0) / 20 * 1e3
    t0 = time.perf_counter()
    for _ in range(20):
        c2 = pairwise_distance(f0, f1, "l2")
        linear_sum_assignment(c2)
    res["end_to_end_ms"] = (time.perf_counter() - t0) / 20 * 1e3
    res["note"] = ("82 MFLOP contraction: launch-latency bound, no roofline claim (SURVEY 8d); l2 = the reference's "
                   "difference form, l2_gemm = norm expansion on fp32 MFMA; assignment on the host like the reference")
    out["config4_matching_200x200x1024"] = res

    # the whole-map reductions the callers run on .data (navigation_policy.py:208-221, agent.py:330-331):
    # pure HBM scans, priced against the bytes they have to read
    from mass_amd.utils.reductions import amax_z, column_occupied
    sem = SemanticProjectionLayer(**lay_kw).to(dev)
    sem.data.normal_()
    occ = OccupancyProjectionLayer(**{k: v for k, v in lay_kw.items() if k != "feature_size"}).to(dev)
    occ.data.uniform_()
    red = {}
    for name, fn, nbytes in (("amax_z_256x256x256x54", lambda: amax_z(sem.data), sem.data.numel() * 4 + MAP * MAP * C * 4),
                             ("column_occupied_occupancy_256^3", lambda: column_occupied(occ.data, None, 0.5), occ.data.numel() * 4 + MAP * MAP)):
        try:
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            red[name] = dict(ms=ms, algorithmic_bytes=nbytes, GBps=nbytes / (ms * 1e-3) / 1e9,
                             frac_of_hbm_peak=nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS)
        except Exception as exc:          # a reduction failing must not lose the headline line
            red[name] = dict(error=repr(exc))
    out["f2_whole_map_reductions"] = red
    return out


# ----------------------------------------------------------------------------------------------
def episode_split(prepared, layers, batch):
    """Where one episode's time goes (ms): the two maps' fusion, then predict_scene_differences cut into the whole-map
    reduction (amax_z), the contour step on the host, the per-box moments, the pairwise cost and the assignment.
    Every piece is bracketed by torch.cuda.synchronize(), so the sum is an upper bound of the pipelined time."""
    import mass_amd.utils.experimentation as ex
    import mass_amd.nn.applications.semantic_projection_layer as spl
    from mass_amd.episodes import run_episode
    acc = {}

    def timed(name, fn):
        def wrapper(*a, **k):
            torch.cuda.synchronize()
            t = time.perf_counter()
            out = fn(*a, **k)
            torch.cuda.synchronize()
            acc[name] = acc.get(name, 0.0) + (time.perf_counter() - t) * 1e3
            return out
        return wrapper
    saved = (spl.amax_z, spl.contour_boxes, spl.lib.mf_roi_moments, ex.pairwise_distance, ex.linear_sum_assignment,
             layers[0].__class__.update_batch, ex.predict_scene_differences)
    try:
        spl.amax_z = timed("amax_z", saved[0])
        spl.contour_boxes = timed("contours_host", saved[1])
        ex.pairwise_distance = timed("pairwise", saved[3])
        ex.linear_sum_assignment = timed("assignment_host", saved[4])
        layers[0].__class__.update_batch = timed("fuse_two_maps", saved[5])
        import mass_amd.episodes as epi
        psd = timed("predict_scene_differences_total", saved[6])
        ex.predict_scene_differences = psd
        torch.cuda.synchronize()
        t = time.perf_counter()
        run_episode(prepared[0], layers, batch=batch)
        torch.cuda.synchronize()
        acc["episode_total"] = (time.perf_counter() - t) * 1e3
    finally:
        spl.amax_z, spl.contour_boxes = saved[0], saved[1]
        ex.pairwise_distance, ex.linear_sum_assignment = saved[3], saved[4]
        layers[0].__class__.update_batch = saved[5]
        ex.predict_scene_differences = saved[6]
    if "predict_scene_differences_total" in acc:
        acc["find_other (roi moments, transfers, python)"] = acc["predict_scene_differences_total"] - sum(
            acc.get(k, 0.0) for k in ("amax_z", "contours_host", "pairwise", "assignment_host"))
    return {k: round(v, 3) for k, v in acc.items()}


def run_episode_rank(args, rank, world, dev, D):
    """BASELINE configs[4]: --episodes synthetic episodes sharded over the ranks like the
    reference shards tasks (agent.py:154-155: episode e on rank e mod world), each a walkthrough and an unshuffle
    semantic map of --episode-frames room frames (seeds 1000 e + phase) fused in sequential 64-frame batches, then
    predict_scene_differences; the per-rank counters are summed by ONE all-reduce at the end (RCCL).  A step is one
    round over the rank's episodes; the frames are resident in HBM before the timed region."""
    from mass_amd.episodes import shard_episodes, prepare_episode, make_episode_layers, run_episode
    mine = shard_episodes(args.episodes, rank, world)
    prepared = [prepare_episode(e, dev, args.episode_frames, H, W, C) for e in mine]
    layers = make_episode_layers(dev, H, W, MAP, C, 0.05)

    def barrier():
        torch.cuda.synchronize()
        D.barrier()
        torch.cuda.synchronize()

    def round_():
        tot = {}
        for p in prepared:
            for k, v in run_episode(p, layers, batch=args.batch).items():
                tot[k] = tot.get(k, 0.0) + v
        return tot
    for _ in range(args.warmup):
        round_()
    barrier()
    t0 = time.perf_counter()
    counters = {}
    for _ in range(args.steps):
        counters = round_()
    barrier()
    wall = time.perf_counter() - t0
    wall_max = D.max_over_ranks(wall)
    split = episode_split(prepared, layers, args.batch) if rank == 0 and prepared else None     # untimed, after the timed region
    for k in ("episodes", "frames", "moved_found", "n_matches", "shift_m", "occupied_voxels", "map_abs_sum"):
        counters.setdefault(k, 0.0)                      # a rank without episodes still takes part in the all-reduce
    metrics = D.reduce_metrics(counters)                 # the run's one data collective
    if rank == 0:
        frames_total = metrics["frames"] * args.steps
        print(json.dumps({
            "metric": "RGB-D frames/s fused into 256^3 semantic voxel map", "value": frames_total / wall_max,
            "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall_max / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"configs[4]: {args.episodes} synthetic episodes (walkthrough + unshuffle semantic map, "
                                   f"{args.episode_frames} room frames each, 480x640 -> 256^3 x 54, sequential batches of "
                                   f"{args.batch}) + predict_scene_differences, episodes sharded over the ranks, one metrics "
                                   f"all-reduce", "episodes": args.episodes, "frames_per_episode": 2 * args.episode_frames},
            "metrics_allreduce": metrics,
            "per_episode_ms": split,
            "host": {"cpu_count": os.cpu_count(), "torch_threads": torch.get_num_threads(),
                     "OMP_NUM_THREADS": os.environ.get("OMP_NUM_THREADS")},
            "note": "strong scaling: the episode count is fixed, ranks share it; includes find() + matching per episode; "
                    "per_episode_ms: one extra untimed round of rank 0's first episode with a device synchronisation around "
                    "every piece (so the pieces do not overlap as they do in the timed rounds)"}),
            flush=True)
    if world > 1:
        D.barrier()
        torch.distributed.destroy_process_group()


def run_rank(args):
    from mass_amd import distributed as D
    # nccl == RCCL on ROCm.  MF_BENCH_BACKEND=gloo is a rehearsal knob for boxes with fewer GPUs
    # than ranks (ranks then share device rank % device_count); the driver never sets it.
    backend = os.environ.get("MF_BENCH_BACKEND", "nccl")
    n_dev = max(torch.cuda.device_count(), 1)
    rank, world, local_rank = D.init_from_env(backend=backend)
    if backend != "nccl":
        local_rank %= n_dev
    args.gpus = world
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from mass_amd.nn.applications.semantic_projection_layer import SemanticProjectionLayer

    lay_kw = dict(camera_height=H, camera_width=W, map_height=MAP, map_width=MAP, map_depth=MAP,
                  feature_size=C, grid_resolution=0.05)
    if args.workload == "episode":
        return run_episode_rank(args, rank, world, dev, D)
    lay = SemanticProjectionLayer(**lay_kw).train().to(dev)
    stride = args.batch if args.rank_seed_stride < 0 else args.rank_seed_stride
    # `rotate` distinct resident batches per rank, taken in turn: 4 x 98 MB of inputs (and 4 x ~0.6 GB of point
    # records in the two workspaces) do not stay in the 256 MB Infinity Cache from one step to the next
    R = max(1, args.rotate)
    batches = []
    for b in range(R):
        fr = make_frames(args.workload, args.batch, seed0=(rank * R + b) * stride)
        batches.append(dict(frames=fr, poses=lay._poses(fr["position"], fr["yaw"], fr["elevation"]),
                            depth=fr["depth"].to(dev).reshape(args.batch, H, W).contiguous(),
                            label=fr["semantic"].to(dev).contiguous()))
    frames = batches[0]["frames"]
    sequential = args.mode == "sequential"

    from mass_amd import _lib
    from mass_amd.utils.projection import fuse_frames, FusePipeline

    pipe = None if args.no_pipeline else FusePipeline(dev)
    counter = [0]

    def step():
        bt = batches[counter[0] % R]
        counter[0] += 1
        if pipe is None:
            fuse_frames(lay.bins_x, lay.bins_y, lay.bins_z, lay.rays, bt["poses"], bt["depth"], bt["label"], lay.data,
                        interpolation_weight=lay.interpolation_weight, sequential=sequential, workspace=lay._workspace)
        else:       # stages this step's batch on the side stream, commits the previous step's
            pipe.submit(lay.bins_x, lay.bins_y, lay.bins_z, lay.rays, bt["poses"], bt["depth"], bt["label"], lay.data,
                        interpolation_weight=lay.interpolation_weight, sequential=sequential)

    def drain():
        if pipe is not None:
            pipe.flush()

    def barrier():
        torch.cuda.synchronize()
        D.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    drain()
    barrier()
    _lib.check(_lib.lib.mf_profile_enable(1))
    n_prof = min(args.steps, 256)            # the library keeps events for 256 calls
    stage_ms = np.zeros((n_prof, 5), np.float32)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for k in range(args.steps):
        step()                     # stage events are recorded on the streams, nothing synchronises here
    drain()                        # every one of the K steps is complete inside the timed region
    ev1.record()
    barrier()
    wall = time.perf_counter() - t0
    for k in range(n_prof):
        _lib.check(_lib.lib.mf_profile_read(k, stage_ms[k].ctypes.data))
    _lib.check(_lib.lib.mf_profile_enable(0))
    gpu_ms = ev0.elapsed_time(ev1)

    wall_max = D.max_over_ranks(wall)

    # untimed, informational: the tile kernel with nothing beside it (one-call steps)
    kernel_name = last_mode(lay, args.batch, pipe.ws[0] if pipe is not None else None) if sequential else "fuse_dense_kernel (merged: one group)"
    alone_ms = None
    if pipe is not None:
        _lib.check(_lib.lib.mf_profile_enable(1))
        for k in range(4):
            bt = batches[k % R]
            fuse_frames(lay.bins_x, lay.bins_y, lay.bins_z, lay.rays, bt["poses"], bt["depth"], bt["label"], lay.data,
                        interpolation_weight=lay.interpolation_weight, sequential=sequential, workspace=lay._workspace)
            torch.cuda.synchronize()
        one = np.zeros((4, 5), np.float32)
        for k in range(4):
            _lib.check(_lib.lib.mf_profile_read(k, one[k].ctypes.data))
        _lib.check(_lib.lib.mf_profile_enable(0))
        alone_ms = float(one[2:, 3].mean())

    # ---- bytes one launch has to move: inputs + union of the touched voxels, read + written once ----
    # (per resident batch; a step's figures are the means over the batches the steps take in turn)
    T, valid_pts, unions = [], 0, []
    for bt in batches:
        Tb, vb, ub = footprints(lay, bt["poses"], bt["depth"])
        T += Tb
        valid_pts += vb
        unions.append(ub)
    valid_pts /= R
    union = float(np.mean(unions))
    input_bytes = args.batch * H * W * (4 + 1)
    tile_bytes = union * C * 4 * 2
    per_frame_model = sum(H * W * (4 + 1) + Tf * C * 4 * 2 for Tf in T) / R       # SURVEY 8(d), informational
    # the one data collective of the run: SUM all-reduce of the per-rank counters (RCCL)
    metrics = D.reduce_metrics(dict(frames=args.batch * args.steps, valid_points=valid_pts * args.steps,
                                    touched_voxels=sum(T) / R * args.steps, union_voxels=union,
                                    map_abs_sum=float(lay.data.abs().sum(dtype=torch.float64))))

    copy_gbps = copy_bandwidth(dev) if rank == 0 else None
    if rank == 0:
        frames_total = args.batch * args.steps * world
        ms_per_step = wall_max / args.steps * 1e3
        st = stage_dict(stage_ms)
        fuse_ms, step_ms = st["fuse_tiles"], st["call"]
        step_gpu_ms = step_ms if args.no_pipeline else gpu_ms / args.steps
        achieved = tile_bytes / (fuse_ms * 1e-3) / 1e9
        traffic, traffic_src = recorded_traffic(f"{args.workload}_{args.mode}_b{args.batch}")
        dist_name = "A (depth 0.5+4.5U, random poses)" if args.workload == "distA" else "B (box room trajectory)"
        out = {
            "metric": "RGB-D frames/s fused into 256^3 semantic voxel map",
            "value": frames_total / wall_max, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"configs[1]: {args.batch} x 480x640 depth + u8 54-class labels -> 256^3 x 54 "
                                   f"fp32 map at 0.05 m, distribution {dist_name}, {args.mode} blend"
                                   f"{' (= %d layer.update calls)' % args.batch if sequential else ''}, one map per GPU",
                       "frames_per_step": args.batch, "map": [MAP, MAP, MAP, C], "mode": args.mode,
                       "resident_batches": R,
                       "issue": ("one mf_fuse_frames call per step" if args.no_pipeline else
                                 "steps pipelined: mf_fuse_frames_stage of step k+1 (bucketing, side stream) overlaps "
                                 "mf_fuse_frames_commit of step k (tile kernels, in order); all K steps complete inside "
                                 "the timed region")},
            "roofline": {"bound": "hbm",
                         "kernel": kernel_name + " (picked on the device from the call's point density)",
                         "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "measured_copy_GBps": copy_gbps, "frac_vs_copy": achieved / copy_gbps,
                         "traffic": traffic,
                         "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": tile_bytes, "kernel_ms": fuse_ms,
                         "kernel_ms_unoverlapped": alone_ms,
                         "frac_unoverlapped": (tile_bytes / (alone_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if alone_ms else None,
                         "union_voxels": union,
                         "traffic_frac_of_peak": (traffic / (fuse_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                         "note": "kernel_ms is measured as the steps are timed (with the next step's bucketing kernels "
                                 "sharing the CUs unless --no-pipeline); "
                                 "algorithmic bytes of the tile kernel = union over the launch's frames of the touched "
                                 "voxels x C x 4 B x (1 read + 1 write): the launch keeps tiles in LDS across its "
                                 "frames, so each touched voxel has to cross HBM once each way; kernel_ms = mean "
                                 "HIP-event time of fuse_tiles over the timed steps (mf_profile_*, on the launch stream)"},
            # a pipelined step's time is the step rate (its stages overlap the neighbours'): GPU time of the whole
            # timed region / K; unpipelined it is the sum of the stages of one call
            "roofline_step": {"achieved": (input_bytes + tile_bytes) / (step_gpu_ms * 1e-3) / 1e9,
                              "frac": (input_bytes + tile_bytes) / (step_gpu_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "algorithmic_bytes_per_step": input_bytes + tile_bytes,
                              "step_ms_gpu": step_gpu_ms, "stage_ms": st},
            "per_frame_model": {"bytes_per_step": per_frame_model,
                                "equivalent_GBps": per_frame_model / (step_gpu_ms * 1e-3) / 1e9,
                                "note": "SURVEY 8(d): sum_f H*W*5 + T_f*C*8, i.e. what B separate layer.update() calls "
                                        "would have to move; informational only (a fused launch moves less), not a "
                                        "roofline fraction"},
            "touched_voxels_per_frame_mean": float(np.mean(T)),
            "gpu_ms_total": gpu_ms,
            "kernel_sources_sha16": sources_sha(),
            "metrics_allreduce": metrics,
        }
        if world == 1 and not args.no_cpu_baseline:
            cb, ref_layer, n_ref = cpu_baseline(frames, args.cpu_frames)
            out["cpu_baseline"] = cb
            if sequential:
                out["parity"] = parity_vs_oracle(lay_kw, frames, n_ref, ref_layer, dev, pipe is not None)
            del ref_layer
        if world == 1 and not args.no_extras:
            del lay
            torch.cuda.empty_cache()
            poses, depth, label = batches[0]["poses"], batches[0]["depth"], batches[0]["label"]
            fa = frames
            if args.workload != "distA":
                fa = make_frames("distA", args.batch, 0)
                probe = SemanticProjectionLayer(**lay_kw).to(dev)
                poses = probe._poses(fa["position"], fa["yaw"], fa["elevation"])
                del probe
                depth = fa["depth"].to(dev).reshape(args.batch, H, W).contiguous()
                label = fa["semantic"].to(dev).contiguous()
            del batches[1:]
            torch.cuda.empty_cache()
            out["other_workloads"] = extra_workloads(lay_kw, dev, fa, poses, depth, label)
        print(json.dumps(out), flush=True)
    if world > 1:
        D.barrier()
        torch.distributed.destroy_process_group()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, argv))
    run_rank(args)


if __name__ == "__main__":
    main()

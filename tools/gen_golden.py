#!/usr/bin/env python3
"""tools/gen_golden.py — generate tests/golden/* from the REFERENCE itself.

Runs only in the build container (needs /root/reference on PYTHONPATH; the
reference never travels).  It imports the reference's torch-only modules

    mass.utils.projection            (/root/reference/mass/utils/projection.py)
    mass.nn.base_projection_layer    (/root/reference/mass/nn/base_projection_layer.py)

feeds them seeded synthetic inputs and stores inputs + outputs as plain
numpy .npz / .json fixtures (data only, loadable with allow_pickle=False).
While doing so it cross-checks the CPU oracle (oracle/massref.py) against the
reference and aborts on any mismatch, which is what pins the oracle.

    python tools/gen_golden.py            # writes tests/golden/
"""
import hashlib
import json
import os
import sys
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
warnings.filterwarnings("ignore")

from mass.utils import projection as refp                      # noqa: E402  (reference)
from mass.nn.base_projection_layer import BaseProjectionLayer  # noqa: E402  (reference)
from oracle import massref as orc                              # noqa: E402  (our oracle)

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(8)


def np32(t):
    # always a copy: layer.data keeps changing after a snapshot is taken
    return np.array(torch.as_tensor(t, dtype=torch.float32).numpy(), dtype=np.float32, order='C', copy=True)


def check_equal(name, a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape or not np.array_equal(a, b, equal_nan=True):
        bad = int((a != b).sum()) if a.shape == b.shape else -1
        raise SystemExit(f"ORACLE MISMATCH (bit-exact expected) in {name}: {bad} elements differ")


def check_close(name, a, b, rtol=1e-6, atol=1e-7):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if a.shape != b.shape or not np.allclose(a, b, rtol=rtol, atol=atol):
        raise SystemExit(f"ORACLE MISMATCH in {name}: max abs {np.abs(a - b).max():.3e}")


POSES = [  # (position xyz_world[3], yaw, elevation, origin (y, x, z))
    ((0.0, 0.0, 0.0), 0.0, 0.0, (0.0, 0.0, 0.0)),
    ((0.1, -0.2, 0.3), 0.7, -0.5, (0.0, 0.0, 0.0)),
    ((0.3, 0.25, 0.1), np.pi / 2, -np.pi / 6, (0.0, 0.0, 0.0)),
    ((1.2, -3.6, 0.2), 0.7, 0.0, (-3.7501, 1.25, 0.123456)),
    ((1.0, -3.9, 0.0), 3.9, -0.5, (-3.7501, 1.25, 0.123456)),
    ((-0.4, 0.4, -0.2), 5.5, 0.3, (0.0, 0.0, 0.0)),
    ((0.0, 0.0, 1.4), 2.2, -1.2, (0.0, 0.0, 0.0)),
    ((1.45, 1.45, 0.0), 0.785, -0.1, (0.0, 0.0, 0.0)),
]
H, W, FOV, MAP, RES = 48, 64, 90.0, 32, 0.1


def ref_layer(C, origin=(0.0, 0.0, 0.0), h=H, w=W, m=MAP, res=RES, iw=0.5, fov=FOV, md=None):
    return BaseProjectionLayer(camera_height=h, camera_width=w, vertical_fov=fov,
                               map_height=m, map_width=m, map_depth=md or m, feature_size=C,
                               origin_y=origin[0], origin_x=origin[1], origin_z=origin[2],
                               grid_resolution=res, interpolation_weight=iw)


def orc_layer(C, origin=(0.0, 0.0, 0.0), h=H, w=W, m=MAP, res=RES, iw=0.5, fov=FOV, md=None):
    return orc.RefProjectionLayer(camera_height=h, camera_width=w, vertical_fov=fov,
                                  map_height=m, map_width=m, map_depth=md or m, feature_size=C,
                                  origin_y=origin[0], origin_x=origin[1], origin_z=origin[2],
                                  grid_resolution=res, interpolation_weight=iw)


# ---------------------------------------------------------------------------
def gen_geom_small():
    g = torch.Generator().manual_seed(1234)
    out = {}
    lay = ref_layer(1)
    out["rays_cam"] = np32(lay.rays)
    check_equal("project_camera_rays", out["rays_cam"], orc_layer(1).rays.numpy())
    pix = torch.arange(H * W, dtype=torch.int64).view(H, W, 1)
    for i, (pos, yaw, el, org) in enumerate(POSES):
        lay = ref_layer(1, org)
        depth = 0.2 + 3.0 * torch.rand(H, W, 1, generator=g)
        yaw_t, el_t = torch.as_tensor(yaw, dtype=torch.float32), torch.as_tensor(el, dtype=torch.float32)
        eye = refp.spherical_to_cartesian(yaw_t, el_t)
        up = refp.spherical_to_cartesian(yaw_t, el_t + np.pi / 2)
        R = torch.stack([torch.cross(eye, up), up, -eye], dim=-1)
        world = refp.transform_rays(lay.rays, eye, up)
        position = torch.as_tensor(pos, dtype=torch.float32)
        i0, i1, i2, r0, r1, r2, pid = refp.bin_rays(lay.bins_x, lay.bins_y, lay.bins_z,
                                                    position, world, depth, pix)
        valid = np.zeros(H * W, np.uint8)
        valid[pid[:, 0].numpy()] = 1
        pre = f"p{i}_"
        out[pre + "position"], out[pre + "yaw"], out[pre + "elevation"] = np32(position), np32(yaw_t), np32(el_t)
        out[pre + "origin_yxz"] = np.asarray(org, np.float64)
        out[pre + "bins_x"], out[pre + "bins_y"], out[pre + "bins_z"] = np32(lay.bins_x), np32(lay.bins_y), np32(lay.bins_z)
        out[pre + "depth"], out[pre + "eye"], out[pre + "up"], out[pre + "R"] = np32(depth), np32(eye), np32(up), np32(R)
        out[pre + "world_rays"] = np32(world)
        out[pre + "valid"] = valid.reshape(H, W)
        for k, v in zip(("ind0", "ind1", "ind2"), (i0, i1, i2)):
            out[pre + k] = v.numpy().astype(np.int64)
        for k, v in zip(("ratio0", "ratio1", "ratio2"), (r0, r1, r2)):
            out[pre + k] = np32(v)
        # ---- oracle cross-check (bit exact) ----
        check_equal("s2c eye", orc.spherical_to_cartesian(yaw_t, el_t).numpy(), out[pre + "eye"])
        check_equal("rotation", orc.rotation_from(eye, up).numpy(), out[pre + "R"])
        ow = orc.transform_rays(lay.rays, eye, up)
        check_equal(f"transform_rays pose {i}", ow.numpy(), out[pre + "world_rays"])
        o = orc.bin_rays(lay.bins_x, lay.bins_y, lay.bins_z, position, ow, depth, pix)
        for k, a in zip(("ind0", "ind1", "ind2", "ratio0", "ratio1", "ratio2"), o[:6]):
            check_equal(f"bin_rays {k} pose {i}", a.numpy(), out[pre + k])
        check_equal(f"bin_rays pix pose {i}", o[6].numpy(), pid.numpy())
        check_equal("bins", orc.make_bins(org[1], MAP, RES).numpy(), out[pre + "bins_x"])
    out["n_poses"] = np.int64(len(POSES))
    np.savez_compressed(os.path.join(OUT, "geom_small.npz"), **out)
    print("geom_small: ok,", sum(int(out[f'p{i}_valid'].sum()) for i in range(len(POSES))), "valid points")


# ---------------------------------------------------------------------------
def frame(g, C, kind):
    depth = 0.2 + 3.0 * torch.rand(H, W, 1, generator=g)
    if kind == "label":
        lab = torch.randint(0, C, (H, W), generator=g)
        feat = torch.nn.functional.one_hot(lab, C).to(torch.float32)
        return depth, feat, lab.numpy().astype(np.int64)
    if kind == "ones":
        return depth, torch.ones_like(depth), None
    return depth, torch.rand(H, W, C, generator=g), None


def gen_splat_small():
    g = torch.Generator().manual_seed(4321)
    out = {}
    cases = [(1, "ones"), (3, "dense"), (5, "dense"), (5, "label")]
    for C, kind in cases:
        tag = f"C{C}{kind}_"
        frames = []
        for j, pi in enumerate((1, 2, 5)):
            depth, feat, lab = frame(g, C, kind)
            frames.append((POSES[pi], depth, feat, lab))
            out[tag + f"f{j}_pose"] = np.int64(pi)
            out[tag + f"f{j}_depth"] = np32(depth)
            if kind == "dense":
                out[tag + f"f{j}_feat"] = np32(feat)
            if kind == "label":
                out[tag + f"f{j}_label"] = lab
        # (i)/(ii): sequential frames on a zero map; store the map after each
        lay, ol = ref_layer(C), orc_layer(C)
        for j, ((pos, yaw, el, _), depth, feat, _) in enumerate(frames):
            obs = dict(position=np.asarray(pos, np.float32), yaw=yaw, elevation=el, depth=depth, features=feat)
            lay.update(obs)
            ol.update(obs)
            out[tag + f"seq{j}_map"] = np32(lay.data)
            check_equal(f"splat {tag} seq{j}", ol.data.numpy(), out[tag + f"seq{j}_map"])
        # (iii): the three frames merged through the functional API (leading batch)
        lay = ref_layer(C)
        ys = torch.tensor([f[0][1] for f in frames], dtype=torch.float32)
        es = torch.tensor([f[0][2] for f in frames], dtype=torch.float32)
        org = torch.tensor([f[0][0] for f in frames], dtype=torch.float32)
        # one reference transform_rays call per frame: with a batch of exactly 3 poses the
        # reference's dim-less torch.cross (projection.py:104) would pick dim 0 (latent defect)
        world = torch.stack([refp.transform_rays(lay.rays, refp.spherical_to_cartesian(ys[b], es[b]),
                                                 refp.spherical_to_cartesian(ys[b], es[b] + np.pi / 2))
                             for b in range(3)])
        depth_b = torch.stack([f[1] for f in frames]); feat_b = torch.stack([f[2] for f in frames])
        ix, iy, iz, rx, ry, rz, fb = refp.bin_rays(lay.bins_x, lay.bins_y, lay.bins_z, org, world, depth_b, feat_b)
        refp.update_feature_map(iy, ix, iz, ry, rx, rz, fb, lay.data, interpolation_weight=0.5)
        out[tag + "merged_map"] = np32(lay.data)
        omap = torch.zeros_like(lay.data)
        wr = torch.stack([orc.transform_rays(lay.rays, orc.spherical_to_cartesian(ys[b], es[b]),
                                             orc.spherical_to_cartesian(ys[b], es[b] + np.pi / 2)) for b in range(3)])
        oix, oiy, oiz, orx, ory, orz, ofb = orc.bin_rays(lay.bins_x, lay.bins_y, lay.bins_z, org, wr, depth_b, feat_b)
        orc.update_feature_map(oiy, oix, oiz, ory, orx, orz, ofb, omap, interpolation_weight=0.5)
        check_equal(f"splat {tag} merged", omap.numpy(), out[tag + "merged_map"])
        # (iv): one frame onto a random non-zero map, iw = 0.3
        init = torch.rand(MAP, MAP, MAP, C, generator=g) * (torch.rand(MAP, MAP, MAP, 1, generator=g) < 0.5)
        out[tag + "init_map"] = np32(init)
        lay, ol = ref_layer(C, iw=0.3), orc_layer(C, iw=0.3)
        lay.data.copy_(init); ol.data.copy_(init)
        (pos, yaw, el, _), depth, feat, _ = frames[0]
        obs = dict(position=np.asarray(pos, np.float32), yaw=yaw, elevation=el, depth=depth, features=feat)
        lay.update(obs); ol.update(obs)
        out[tag + "onto_map"] = np32(lay.data)
        check_equal(f"splat {tag} onto", ol.data.numpy(), out[tag + "onto_map"])
    np.savez_compressed(os.path.join(OUT, "splat_small.npz"), **out)
    print("splat_small: ok")


# ---------------------------------------------------------------------------
def gen_edge_cases():
    """Functional API with hand-made rays: depths {0, 10, 10.0001, nan, inf, <0},
    points exactly on bin edges, ratios of exactly 0.5, border voxels, out of map."""
    lay = ref_layer(2, origin=(0.05, -0.05, 0.0))
    bx, by, bz = lay.bins_x, lay.bins_y, lay.bins_z
    rows = []
    origin = torch.tensor([0.0, 0.0, 0.0])
    X = torch.tensor([1.0, 0.0, 0.0]); Y = torch.tensor([0.0, 1.0, 0.0]); Z = torch.tensor([0.0, 0.0, 1.0])
    for d in (0.0, 10.0, 10.0001, float("nan"), float("inf"), -0.5, -0.0, 1e-30, 9.999999):
        rows.append((X * 0.1, d))
    for e in (0, 1, 5, 16, 31, 32):                      # exactly on edges (x, y, z axes)
        rows.append((X, float(bx[e]))); rows.append((Y, float(by[e]))); rows.append((Z, float(bz[e])))
        rows.append((-X, -float(bx[e]))); rows.append((X, float(np.nextafter(np.float32(bx[e]), np.float32(-9)))))
    for e in (0, 7, 31):                                  # centre of a voxel: ratio 0.5 (or next to it)
        c = (float(bx[e]) + float(bx[e + 1])) / 2
        rows.append((X, c)); rows.append((X, float(np.nextafter(np.float32(c), np.float32(9)))))
    diag = torch.tensor([1.0, 1.0, 1.0])
    for d in (0.01, 0.7, 1.55, 1.6, 1.649, 1.7, 3.0):    # towards the +++ corner and beyond
        rows.append((diag, d)); rows.append((-diag, d)); rows.append((torch.tensor([1.0, -1.0, 0.3]), d))
    rays = torch.stack([r for r, _ in rows]).view(1, -1, 3)
    depth = torch.tensor([d for _, d in rows], dtype=torch.float32).view(1, -1, 1)
    n = rays.shape[1]
    pid = torch.arange(n).view(1, n, 1)
    feat = torch.stack([torch.linspace(0.1, 1.0, n), torch.ones(n)], dim=-1).view(1, n, 2)
    i0, i1, i2, r0, r1, r2, p, f = refp.bin_rays(bx, by, bz, origin, rays, depth, pid, feat)
    valid = np.zeros(n, np.uint8); valid[p[:, 0].numpy()] = 1
    init = torch.full((MAP, MAP, MAP, 2), 0.25)
    m = init.clone()
    refp.update_feature_map(i1, i0, i2, r1, r0, r2, f, m, interpolation_weight=0.5)
    out = dict(bins_x=np32(bx), bins_y=np32(by), bins_z=np32(bz), origin=np32(origin), rays=np32(rays),
               depth=np32(depth), feat=np32(feat), valid=valid, ind0=i0.numpy(), ind1=i1.numpy(), ind2=i2.numpy(),
               ratio0=np32(r0), ratio1=np32(r1), ratio2=np32(r2), init_value=np.float32(0.25), map_after=np32(m))
    o = orc.bin_rays(bx, by, bz, origin, rays, depth, pid, feat)
    for k, a in zip(("ind0", "ind1", "ind2", "ratio0", "ratio1", "ratio2"), o[:6]):
        check_equal(f"edge {k}", a.numpy(), out[k])
    check_equal("edge pid", o[6].numpy(), p.numpy())
    om = init.clone()
    orc.update_feature_map(o[1], o[0], o[2], o[4], o[3], o[5], o[7], om, interpolation_weight=0.5)
    check_equal("edge map", om.numpy(), out["map_after"])
    np.savez_compressed(os.path.join(OUT, "edge_cases.npz"), **out)
    print(f"edge_cases: ok, {int(valid.sum())}/{n} valid")


# ---------------------------------------------------------------------------
def digest_of(layer_data, flat_ids):
    d = layer_data
    ids = np.sort(np.asarray(flat_ids, np.int64))
    return dict(occupied=int((d != 0).any(-1).sum()), nonzero=int((d != 0).sum()),
                sum=float(d.double().sum()), max=float(d.max()),
                channel_sums=[float(x) for x in d.double().sum(dim=(0, 1, 2))],
                n_valid=int(ids.size), sha256_sorted_flat_ids=hashlib.sha256(ids.tobytes()).hexdigest())


def gen_digest_480x640():
    """inputs are regenerated from the seed on
    both sides, only digests are stored."""
    Hh, Ww, C = 480, 640, 54
    res = {}
    for m in (128, 256):
        g = torch.Generator().manual_seed(0)
        depth = 0.5 + 2.5 * torch.rand(Hh, Ww, 1, generator=g)
        label = torch.randint(0, C, (Hh, Ww), generator=g)
        feat = torch.nn.functional.one_hot(label, C).to(torch.float32)
        lay = ref_layer(C, h=Hh, w=Ww, m=m, res=0.05)
        obs = dict(position=np.asarray((0.1, -0.2, 0.3), np.float32), yaw=0.7, elevation=-0.5,
                   depth=depth, features=feat)
        world = refp.transform_rays(lay.rays, refp.spherical_to_cartesian(torch.tensor(0.7), torch.tensor(-0.5)),
                                    refp.spherical_to_cartesian(torch.tensor(0.7), torch.tensor(-0.5) + np.pi / 2))
        ix, iy, iz, *_ = refp.bin_rays(lay.bins_x, lay.bins_y, lay.bins_z,
                                       torch.tensor((0.1, -0.2, 0.3)), world, depth)
        flat = ((iy * m + ix) * m + iz).numpy()
        entry = {}
        ol = orc_layer(C, h=Hh, w=Ww, m=m, res=0.05) if m == 128 else None
        for rep in range(3 if m == 128 else 1):
            lay.update(obs)
            entry[f"after_{rep + 1}"] = digest_of(lay.data, flat)
            if ol is not None:
                ol.update(obs)
                check_equal(f"digest m={m} rep={rep}", ol.data.numpy(), lay.data.numpy())
        entry["depth_sha256"] = hashlib.sha256(np32(depth).tobytes()).hexdigest()
        entry["label_sha256"] = hashlib.sha256(label.numpy().astype(np.int64).tobytes()).hexdigest()
        res[f"map{m}"] = entry
        print(f"digest m={m}:", {k: v for k, v in entry[f'after_{3 if m == 128 else 1}'].items() if k in ('occupied', 'sum', 'max')})
    res["recipe"] = ("torch.Generator().manual_seed(0); depth = 0.5 + 2.5*rand(480,640,1); label = randint(0,54,(480,640)); "
                     "pos (0.1,-0.2,0.3) yaw 0.7 el -0.5; fov 90; res 0.05; iw 0.5; same frame applied N times")
    res["torch_version"] = torch.__version__
    with open(os.path.join(OUT, "digest_480x640.json"), "w") as f:
        json.dump(res, f, indent=1)


# ---------------------------------------------------------------------------
def gen_match_small():
    """experimentation.py:261-265,284-287 restated with the same two library
    calls (the module itself cannot be imported: ai2thor/rearrange absent)."""
    from scipy.optimize import linear_sum_assignment
    import scipy
    out = {"scipy_version": np.array(scipy.__version__)}
    g = torch.Generator().manual_seed(77)
    cases = [(1, 1, 3), (3, 5, 3), (5, 3, 256), (5, 5, 256), (40, 40, 256), (1, 4, 1024), (200, 200, 256)]
    for n0, n1, d in cases:
        f0 = torch.randn(n0, d, generator=g)
        f1 = torch.randn(n1, d, generator=g)
        if n0 == n1 and n0 >= 5:     # realistic: same objects seen twice, permuted, small perturbation
            f1 = f0[torch.randperm(n0, generator=g)] + 0.05 * torch.randn(n0, d, generator=g)
        cost = torch.linalg.norm(f0.unsqueeze(1) - f1.unsqueeze(0), dim=2)
        rows, cols = linear_sum_assignment(cost.detach().cpu().numpy())
        tag = f"n{n0}x{n1}d{d}_"
        out[tag + "f0"], out[tag + "f1"], out[tag + "cost"] = np32(f0), np32(f1), np32(cost)
        out[tag + "rows"], out[tag + "cols"] = rows.astype(np.int64), cols.astype(np.int64)
        check_close(f"pairwise {tag}", orc.pairwise_l2(f0, f1).numpy(), out[tag + "cost"], rtol=2e-6, atol=1e-6)
    # tie-heavy integer costs pin the solver's tie-breaking
    for n0, n1 in ((6, 6), (4, 7), (7, 4)):
        cost = torch.randint(0, 3, (n0, n1), generator=g).to(torch.float32)
        rows, cols = linear_sum_assignment(cost.numpy())
        tag = f"ties{n0}x{n1}_"
        out[tag + "cost"], out[tag + "rows"], out[tag + "cols"] = np32(cost), rows.astype(np.int64), cols.astype(np.int64)
    # config 4 (200 x 200 x 1024, seeds 0/1): inputs regenerated from seed, digest + result stored
    f0 = torch.randn(200, 1024, generator=torch.Generator().manual_seed(0))
    f1 = torch.randn(200, 1024, generator=torch.Generator().manual_seed(1))
    cost = torch.linalg.norm(f0.unsqueeze(1) - f1.unsqueeze(0), dim=2)
    rows, cols = linear_sum_assignment(cost.numpy())
    out["cfg4_input_sha256"] = np.array(hashlib.sha256(np32(f0).tobytes() + np32(f1).tobytes()).hexdigest())
    out["cfg4_cost"], out["cfg4_rows"], out["cfg4_cols"] = np32(cost), rows.astype(np.int64), cols.astype(np.int64)
    check_close("pairwise cfg4", orc.pairwise_l2(f0, f1).numpy(), out["cfg4_cost"], rtol=2e-6, atol=1e-6)
    np.savez_compressed(os.path.join(OUT, "match_small.npz"), **out)
    print("match_small: ok")




# ---------------------------------------------------------------------------
def gen_transforms():
    """coordinate transforms, top_down, and the whole-map reductions the
    callers run on the maps (navigation_policy.py:208-221 navigable_area, agent.py:330 amax)."""
    import torch.nn.functional as functional
    g = torch.Generator().manual_seed(99)
    out = {}
    lay = ref_layer(4, origin=(-3.7501, 1.25, 0.123456), m=24, md=12)
    lay.data.copy_(torch.rand(24, 24, 12, 4, generator=g) * (torch.rand(24, 24, 12, 1, generator=g) < 0.15))
    out["origin_yxz"] = np.asarray((-3.7501, 1.25, 0.123456), np.float64)
    out["data"] = np32(lay.data)
    world = torch.cat([torch.rand(200, 3, generator=g) * 4 - 2 + torch.tensor([1.25, -3.7501, 0.12]),
                       torch.tensor([[1.25, -3.7501, 0.123456], [100.0, 100.0, 100.0], [-100.0, -100.0, -100.0]]),
                       torch.stack([lay.bins_x[3], lay.bins_y[5], lay.bins_z[7]]).view(1, 3)])
    out["world"] = np32(world)
    out["clamp_to_world"] = np32(lay.clamp_to_world(world))
    out["world_to_map"] = lay.world_to_map(world).numpy().astype(np.int64)
    out["world_to_map_xy"] = lay.world_to_map(world[:, :2]).numpy().astype(np.int64)
    mapc = torch.cat([torch.rand(200, 3, generator=g) * torch.tensor([26.0, 26.0, 14.0]) - 1.0,
                      torch.tensor([[0.0, 0.0, 0.0], [23.0, 23.0, 11.0], [5.5, 7.25, 3.0]])])
    out["map_coords"] = np32(mapc)
    out["clamp_to_map"] = np32(lay.clamp_to_map(mapc))
    out["map_to_world"] = np32(lay.map_to_world(mapc))
    out["top_down_0_8"] = np32(lay.top_down(depth_slice=slice(0, 8)))
    out["top_down_all"] = np32(lay.top_down(depth_slice=None))
    out["visualize"] = np.asarray(lay.visualize({}, depth_slice=slice(0, 8)), np.float32)
    # navigable_area body (navigation_policy.py:208-221) with padding 2, slice(2, 9), thresholds 0 and 0.5
    for thr in (0.0, 0.5):
        nav = torch.norm(lay.data, p=1, dim=3) > thr
        nav = nav[:, :, slice(2, 9)]
        nav = torch.logical_not(nav.any(dim=2)).to(dtype=lay.data.dtype)
        nav = 1 - functional.max_pool2d(1 - nav.unsqueeze(0), 2 * 2 + 1, stride=1, padding=2).squeeze(0)
        out[f"navigable_thr{thr}"] = np32(nav)
    out["amax_z"] = np32(lay.data.amax(dim=2))          # agent.py:330-331
    np.savez_compressed(os.path.join(OUT, "transforms_small.npz"), **out)
    print("transforms_small: ok")


if __name__ == "__main__":
    which = sys.argv[1:] or ["geom", "splat", "edge", "match", "transforms", "digest"]
    if "geom" in which: gen_geom_small()
    if "splat" in which: gen_splat_small()
    if "edge" in which: gen_edge_cases()
    if "match" in which: gen_match_small()
    if "transforms" in which: gen_transforms()
    if "digest" in which: gen_digest_480x640()
    print("all fixtures written to", OUT)

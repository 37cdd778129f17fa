"""The CPU oracle (oracle/massref.c) against the fixtures the reference itself
produced (tools/gen_golden.py).  Bit-exact: this is what pins the oracle."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

from conftest import SMALL, POSE_OF_FRAME, GOLDEN
from oracle import massref as orc

H, W, MAP, RES = SMALL["H"], SMALL["W"], SMALL["MAP"], SMALL["RES"]


def layer(C, origin=(0.0, 0.0, 0.0), iw=0.5, **kw):
    args = dict(camera_height=H, camera_width=W, vertical_fov=90.0, map_height=MAP, map_width=MAP,
                map_depth=MAP, feature_size=C, origin_y=origin[0], origin_x=origin[1],
                origin_z=origin[2], grid_resolution=RES, interpolation_weight=iw)
    args.update(kw)
    return orc.RefProjectionLayer(**args)


def test_camera_rays(geom):
    assert np.array_equal(layer(1).rays.numpy(), geom["rays_cam"])


@pytest.mark.parametrize("i", range(8))
def test_geometry_pose(geom, i):
    p = f"p{i}_"
    org = geom[p + "origin_yxz"]
    lay = layer(1, origin=tuple(org))
    for ax in "xyz":
        assert np.array_equal(getattr(lay, "bins_" + ax).numpy(), geom[p + "bins_" + ax])
    yaw, el = torch.tensor(geom[p + "yaw"]).reshape(()), torch.tensor(geom[p + "elevation"]).reshape(())
    eye = orc.spherical_to_cartesian(yaw, el)
    up = orc.spherical_to_cartesian(yaw, el + np.pi / 2)
    assert np.array_equal(eye.numpy(), geom[p + "eye"])
    assert np.array_equal(up.numpy(), geom[p + "up"])
    assert np.array_equal(orc.rotation_from(eye, up).numpy(), geom[p + "R"])
    world = orc.transform_rays(lay.rays, eye, up)
    assert np.array_equal(world.numpy(), geom[p + "world_rays"])
    o = orc.bin_rays_dense(lay.bins_x, lay.bins_y, lay.bins_z, torch.tensor(geom[p + "position"]),
                           world, torch.tensor(geom[p + "depth"]))
    valid = o["valid"].astype(bool)
    assert np.array_equal(o["valid"], geom[p + "valid"])
    for k in ("ind0", "ind1", "ind2", "ratio0", "ratio1", "ratio2"):
        assert np.array_equal(o[k][valid], geom[p + k]), k


CASES = [(1, "ones"), (3, "dense"), (5, "dense"), (5, "label")]


def frame_inputs(splat, geom, C, kind, j):
    tag = f"C{C}{kind}_"
    pi = int(splat[tag + f"f{j}_pose"])
    assert pi == POSE_OF_FRAME[j]
    depth = torch.tensor(splat[tag + f"f{j}_depth"])
    if kind == "ones":
        feat = torch.ones_like(depth)
    elif kind == "dense":
        feat = torch.tensor(splat[tag + f"f{j}_feat"])
    else:
        feat = torch.nn.functional.one_hot(torch.tensor(splat[tag + f"f{j}_label"]), C).float()
    return dict(position=geom[f"p{pi}_position"], yaw=geom[f"p{pi}_yaw"].reshape(()),
                elevation=geom[f"p{pi}_elevation"].reshape(()), depth=depth, features=feat)


@pytest.mark.parametrize("C,kind", CASES)
def test_splat_sequential(splat, geom, C, kind):
    lay = layer(C)
    for j in range(3):
        lay.update(frame_inputs(splat, geom, C, kind, j))
        assert np.array_equal(lay.data.numpy(), splat[f"C{C}{kind}_seq{j}_map"]), f"frame {j}"


@pytest.mark.parametrize("C,kind", CASES)
def test_splat_merged(splat, geom, C, kind):
    lay = layer(C)
    obs = [frame_inputs(splat, geom, C, kind, j) for j in range(3)]
    world, org = [], []
    for o in obs:
        yaw, el = torch.tensor(o["yaw"]), torch.tensor(o["elevation"])
        world.append(orc.transform_rays(lay.rays, orc.spherical_to_cartesian(yaw, el),
                                        orc.spherical_to_cartesian(yaw, el + np.pi / 2)))
        org.append(torch.tensor(o["position"]))
    ix, iy, iz, rx, ry, rz, f = orc.bin_rays(lay.bins_x, lay.bins_y, lay.bins_z, torch.stack(org),
                                             torch.stack(world), torch.stack([o["depth"] for o in obs]),
                                             torch.stack([o["features"] for o in obs]))
    orc.update_feature_map(iy, ix, iz, ry, rx, rz, f, lay.data, interpolation_weight=0.5)
    assert np.array_equal(lay.data.numpy(), splat[f"C{C}{kind}_merged_map"])


@pytest.mark.parametrize("C,kind", CASES)
def test_splat_onto_nonzero_map(splat, geom, C, kind):
    lay = layer(C, iw=0.3)
    lay.data.copy_(torch.tensor(splat[f"C{C}{kind}_init_map"]))
    lay.update(frame_inputs(splat, geom, C, kind, 0))
    assert np.array_equal(lay.data.numpy(), splat[f"C{C}{kind}_onto_map"])


def test_edge_cases(edge):
    n = edge["rays"].shape[1]
    pid = torch.arange(n).view(1, n, 1)
    o = orc.bin_rays(torch.tensor(edge["bins_x"]), torch.tensor(edge["bins_y"]), torch.tensor(edge["bins_z"]),
                     torch.tensor(edge["origin"]), torch.tensor(edge["rays"]), torch.tensor(edge["depth"]),
                     pid, torch.tensor(edge["feat"]))
    valid = np.zeros(n, np.uint8)
    valid[o[6][:, 0].numpy()] = 1
    assert np.array_equal(valid, edge["valid"])
    for k, a in zip(("ind0", "ind1", "ind2", "ratio0", "ratio1", "ratio2"), o[:6]):
        assert np.array_equal(a.numpy(), edge[k]), k
    m = torch.full((MAP, MAP, MAP, 2), float(edge["init_value"]))
    orc.update_feature_map(o[1], o[0], o[2], o[4], o[3], o[5], o[7], m, interpolation_weight=0.5)
    assert np.array_equal(m.numpy(), edge["map_after"])


def test_empty_input():
    m = torch.zeros(4, 4, 4, 2)
    e = torch.zeros(0, dtype=torch.int64)
    assert orc.update_feature_map(e, e, e, e.float(), e.float(), e.float(), torch.zeros(0, 2), m) == 0
    assert not m.any()


def test_pairwise_l2(matchfx):
    tags = sorted({k[:-2] for k in matchfx.files if k.endswith("_f0")})
    assert tags
    for t in tags:
        got = orc.pairwise_l2(matchfx[t + "f0"], matchfx[t + "f1"]).numpy()
        np.testing.assert_allclose(got, matchfx[t + "cost"], rtol=2e-6, atol=1e-6)


def test_config1_digest():
    """SURVEY 8(d) config 1 (480x640, 54 classes, 128^3, same frame 3x):
    inputs regenerated from the seed, digests recorded from the reference."""
    with open(os.path.join(GOLDEN, "digest_480x640.json")) as f:
        ref = json.load(f)["map128"]
    g = torch.Generator().manual_seed(0)
    depth = 0.5 + 2.5 * torch.rand(480, 640, 1, generator=g)
    label = torch.randint(0, 54, (480, 640), generator=g)
    if hashlib.sha256(depth.numpy().tobytes()).hexdigest() != ref["depth_sha256"]:
        pytest.skip("torch RNG stream differs from the one the fixture was recorded with")
    feat = torch.nn.functional.one_hot(label, 54).float()
    lay = orc.RefProjectionLayer(camera_height=480, camera_width=640, map_height=128, map_width=128,
                                 map_depth=128, feature_size=54, grid_resolution=0.05)
    obs = dict(position=np.asarray((0.1, -0.2, 0.3), np.float32), yaw=0.7, elevation=-0.5, depth=depth,
               features=feat)
    for rep in range(3):
        lay.update(obs)
        want = ref[f"after_{rep + 1}"]
        d = lay.data
        assert int((d != 0).any(-1).sum()) == want["occupied"]
        assert int((d != 0).sum()) == want["nonzero"]
        assert float(d.double().sum()) == pytest.approx(want["sum"], rel=1e-12)
        assert float(d.max()) == want["max"]
    assert want["occupied"] == 296167      # SURVEY / BASELINE.md anchor


def test_threaded_blend_is_bit_identical_to_the_single_thread():
    """oracle/massref.c may run its blend loop on several threads (each owns a fixed subset of the touched
    voxels and keeps the single-thread visiting order): same bits for 1, 3 and 8 threads, on a map that is
    not zero and with points that pile onto the same voxels."""
    g = torch.Generator().manual_seed(4)
    kw = dict(camera_height=60, camera_width=80, map_height=40, map_width=40, map_depth=24, feature_size=7,
              grid_resolution=0.1)
    init = torch.rand(40, 40, 24, 7, generator=g)
    obs = [dict(position=0.2 * torch.randn(3, generator=g), yaw=float(6.28 * torch.rand((), generator=g)),
                elevation=-0.4, depth=0.9 + 0.1 * torch.rand(60, 80, 1, generator=g),
                features=torch.rand(60, 80, 7, generator=g)) for _ in range(3)]
    before = orc.get_threads()
    outs = []
    try:
        orc.force_threads(True)
        for n in (1, 3, 8):
            orc.set_threads(n)
            lay = orc.RefProjectionLayer(**kw)
            lay.data.copy_(init)
            for o in obs:
                lay.update(o)
            assert orc.last_threads() == n
            outs.append(lay.data.clone())
    finally:
        orc.force_threads(False)
        orc.set_threads(before)
    assert bool((outs[0] != init).any())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])

"""HIP geometry (transform_rays, bin_rays, fused unproject+bin) against the
fixtures recorded from the reference: integer outputs AND fp32 ratios bit-exact."""
import numpy as np
import pytest
import torch

from conftest import SMALL

pytestmark = pytest.mark.gpu
H, W = SMALL["H"], SMALL["W"]


def dev(a, device, dtype=None):
    t = torch.as_tensor(np.asarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(device)


def pose_rows(geom, ids, device):
    from mass_amd.utils.projection import pack_poses
    pos = torch.tensor(np.stack([geom[f"p{i}_position"] for i in ids]))
    eye = torch.tensor(np.stack([geom[f"p{i}_eye"] for i in ids]))
    up = torch.tensor(np.stack([geom[f"p{i}_up"] for i in ids]))
    return pack_poses(pos, eye, up).to(device)


def test_host_pose_math_matches_reference(geom):
    """a1 + rotation stay on the host in torch: bit-identical to the fixtures."""
    from mass_amd.utils.projection import spherical_to_cartesian, rotation_matrix
    for i in range(8):
        yaw = torch.tensor(geom[f"p{i}_yaw"]).reshape(())
        el = torch.tensor(geom[f"p{i}_elevation"]).reshape(())
        eye = spherical_to_cartesian(yaw, el)
        up = spherical_to_cartesian(yaw, el + np.pi / 2)
        assert np.array_equal(eye.numpy(), geom[f"p{i}_eye"])
        assert np.array_equal(up.numpy(), geom[f"p{i}_up"])
        assert np.array_equal(rotation_matrix(eye, up).numpy(), geom[f"p{i}_R"])


def test_transform_rays_bit_exact(geom, device):
    from mass_amd.utils.projection import transform_rays
    cam = dev(geom["rays_cam"], device)
    for i in range(8):
        out = transform_rays(cam, dev(geom[f"p{i}_eye"], device), dev(geom[f"p{i}_up"], device))
        assert np.array_equal(out.cpu().numpy(), geom[f"p{i}_world_rays"]), f"pose {i}"
    eyes = dev(np.stack([geom[f"p{i}_eye"] for i in range(8)]), device)
    ups = dev(np.stack([geom[f"p{i}_up"] for i in range(8)]), device)
    out = transform_rays(cam, eyes, ups).cpu().numpy()
    for i in range(8):
        assert np.array_equal(out[i], geom[f"p{i}_world_rays"])


@pytest.mark.parametrize("i", range(8))
def test_bin_rays_bit_exact(geom, device, i):
    from mass_amd.utils.projection import bin_rays, bin_rays_dense
    p = f"p{i}_"
    args = [dev(geom[p + k], device) for k in ("bins_x", "bins_y", "bins_z", "position", "world_rays", "depth")]
    dense = bin_rays_dense(*args)
    assert np.array_equal(dense[6].cpu().numpy(), geom[p + "valid"])
    pix = torch.arange(H * W, device=device).view(H, W, 1)
    out = bin_rays(*args, pix)
    for k, a in zip(("ind0", "ind1", "ind2", "ratio0", "ratio1", "ratio2"), out[:6]):
        assert a.dtype == (torch.int64 if k.startswith("ind") else torch.float32)
        assert np.array_equal(a.cpu().numpy(), geom[p + k]), k
    assert np.array_equal(out[6][:, 0].cpu().numpy(), np.nonzero(geom[p + "valid"].reshape(-1))[0])


def test_fused_unproject_bin_equals_two_step(geom, device):
    """The hot path's fused a3+a4 (camera rays + pose rows) gives the same
    bits as transform_rays followed by bin_rays, i.e. as the fixtures."""
    from mass_amd.utils.projection import unproject_bin
    cam = dev(geom["rays_cam"], device)
    for group in ([0, 1, 2, 5, 6, 7], [3, 4]):          # same map origin inside a group
        p0 = f"p{group[0]}_"
        bins = [dev(geom[p0 + k], device) for k in ("bins_x", "bins_y", "bins_z")]
        depth = dev(np.stack([geom[f"p{i}_depth"] for i in group]), device)
        out = unproject_bin(*bins, cam, pose_rows(geom, group, device), depth)
        for n, i in enumerate(group):
            p = f"p{i}_"
            valid = geom[p + "valid"].astype(bool)
            assert np.array_equal(out[6][n].cpu().numpy().astype(bool), valid)
            for k, a in zip(("ind0", "ind1", "ind2", "ratio0", "ratio1", "ratio2"), out[:6]):
                assert np.array_equal(a[n].cpu().numpy()[valid], geom[p + k]), (i, k)


def test_edge_cases_bit_exact(edge, device):
    """depth in {0, 10, 10.0001, nan, inf, <0}, points exactly on bin edges,
    ratio exactly 0.5, border voxels, rays leaving the map."""
    from mass_amd.utils.projection import bin_rays
    n = edge["rays"].shape[1]
    pid = torch.arange(n, device=device).view(1, n, 1)
    out = bin_rays(dev(edge["bins_x"], device), dev(edge["bins_y"], device), dev(edge["bins_z"], device),
                   dev(edge["origin"], device), dev(edge["rays"], device), dev(edge["depth"], device), pid)
    valid = np.zeros(n, np.uint8)
    valid[out[6][:, 0].cpu().numpy()] = 1
    assert np.array_equal(valid, edge["valid"])
    for k, a in zip(("ind0", "ind1", "ind2", "ratio0", "ratio1", "ratio2"), out[:6]):
        assert np.array_equal(a.cpu().numpy(), edge[k]), k


def test_nonuniform_and_tiny_bins(device):
    """The edge estimate is only a guess: non-uniform edges must still give
    torch.bucketize(right=True) - 1."""
    from mass_amd.utils.projection import bin_rays_dense
    g = torch.Generator().manual_seed(3)
    edges = torch.cumsum(torch.rand(40, generator=g) ** 3 + 1e-3, 0) - 2.0
    pts = (torch.rand(1, 5000, 1, generator=g) * 8 - 3)
    rays = torch.tensor([1.0, 0.0, 0.0]).expand(1, 5000, 3).contiguous()
    two = torch.tensor([-100.0, 100.0])
    out = bin_rays_dense(edges.to(device), two.to(device), two.to(device), torch.zeros(3, device=device),
                         rays.to(device), pts.to(device), min_ray_depth=-100.0, max_ray_depth=100.0)
    want = torch.bucketize(pts[0, :, 0], edges, right=True) - 1
    assert np.array_equal(out[0][0].cpu().numpy(), want.numpy())
    ok = (want >= 0) & (want < edges.numel() - 1)
    assert np.array_equal(out[6][0].cpu().numpy().astype(bool), ok.numpy())


@pytest.mark.parametrize("seed", range(12))
def test_random_configurations_bit_exact_vs_oracle(device, seed):
    """Random cameras, map sizes, resolutions, origins and poses: the fused unproject+bin of the
    hot path must equal the oracle (itself bit-identical to the reference) in every index, ratio
    and validity bit."""
    from mass_amd.nn.base_projection_layer import BaseProjectionLayer
    from mass_amd.utils.projection import unproject_bin
    from oracle import massref as orc
    g = torch.Generator().manual_seed(1000 + seed)
    r = lambda lo, hi: float(lo + (hi - lo) * torch.rand((), generator=g))
    h, w = int(torch.randint(8, 97, (1,), generator=g)), int(torch.randint(8, 129, (1,), generator=g))
    kw = dict(camera_height=h, camera_width=w, vertical_fov=r(50, 110), map_height=int(torch.randint(5, 70, (1,), generator=g)),
              map_width=int(torch.randint(5, 70, (1,), generator=g)), map_depth=int(torch.randint(3, 40, (1,), generator=g)),
              feature_size=1, grid_resolution=r(0.02, 0.3), origin_y=r(-5, 5), origin_x=r(-5, 5), origin_z=r(-1, 1))
    lay = BaseProjectionLayer(**kw).to(device)
    ol = orc.RefProjectionLayer(**kw)
    assert np.array_equal(lay.rays.cpu().numpy(), ol.rays.numpy())
    n = 3
    pos = torch.tensor([kw["origin_x"], kw["origin_y"], kw["origin_z"]]) + 0.5 * torch.randn(n, 3, generator=g)
    yaw, el = 6.3 * torch.rand(n, generator=g), 1.5 * torch.rand(n, generator=g) - 1.0
    span = kw["grid_resolution"] * max(kw["map_height"], kw["map_width"])
    depth = span * torch.rand(n, h, w, 1, generator=g) ** 2
    out = unproject_bin(lay.bins_x, lay.bins_y, lay.bins_z, lay.rays, lay._poses(pos, yaw, el), depth.to(device))
    n_valid = 0
    for b in range(n):
        world = orc.transform_rays(ol.rays, orc.spherical_to_cartesian(yaw[b], el[b]),
                                   orc.spherical_to_cartesian(yaw[b], el[b] + np.pi / 2))
        o = orc.bin_rays_dense(ol.bins_x, ol.bins_y, ol.bins_z, pos[b], world, depth[b])
        valid = o["valid"].astype(bool)
        assert np.array_equal(out[6][b].cpu().numpy().astype(bool), valid)
        for k, a in zip(("ind0", "ind1", "ind2", "ratio0", "ratio1", "ratio2"), out[:6]):
            assert np.array_equal(a[b].cpu().numpy()[valid], o[k][valid]), (seed, b, k)
        n_valid += int(valid.sum())
    assert n_valid > 0 or seed == 10          # seed 10 draws three cameras that see nothing of the map

"""BASELINE config 5 in miniature: one episode = a walkthrough map and an unshuffle map built
from two synthetic trajectories of the same room in which one object class has moved, then
predict_scene_differences (find + pairwise cost + assignment).  Every stage is checked against
the CPU oracle; the counters an episode contributes to the multi-GPU all-reduce are computed."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
H, W, M, MD, C = 60, 80, 64, 32, 8


def scene(moved, n=10, seed=0):
    """Room trajectory whose labels carry two box-shaped 'objects': class 3 fixed, class 5 at a
    position that depends on `moved`; everything else is background class 0."""
    from mass_amd.episodes import room_trajectory
    tr = room_trajectory(n, H, W, seed=seed, num_classes=C)
    from mass_amd.utils.projection import project_camera_rays, spherical_to_cartesian, rotation_matrix
    cam = project_camera_rays(H, W, H / 2.0, H / 2.0)
    sem = torch.zeros(n, H, W, dtype=torch.int64)
    for t in range(n):
        eye = spherical_to_cartesian(tr["yaw"][t], tr["elevation"][t])
        up = spherical_to_cartesian(tr["yaw"][t], tr["elevation"][t] + np.pi / 2)
        q = (cam.unsqueeze(-2) * rotation_matrix(eye, up)).sum(-1)
        hit = tr["position"][t] + q * tr["depth"][t]
        for cls, centre in ((3, (2.95, 0.5, -0.8)), (5, (-1.0, 2.95, -0.9) if not moved else (1.2, 2.95, -0.9))):
            d = (hit - torch.tensor(centre)).abs()
            sem[t][(d[..., 0] < 0.45) & (d[..., 1] < 0.45) & (d[..., 2] < 0.45)] = cls
    tr["semantic"] = sem
    return tr


def build(device, tr):
    from mass_amd.nn.applications.semantic_projection_layer import SemanticProjectionLayer
    from mass_amd.nn.base_projection_layer import BaseProjectionLayer
    from oracle import massref as orc
    kw = dict(camera_height=H, camera_width=W, map_height=M, map_width=M, map_depth=MD, grid_resolution=0.12)
    sem = SemanticProjectionLayer(feature_size=C, **kw).to(device)
    rgb = BaseProjectionLayer(feature_size=3, **kw).to(device)
    o_sem = orc.RefProjectionLayer(feature_size=C, **kw)
    o_rgb = orc.RefProjectionLayer(feature_size=3, **kw)
    sem.update_batch(dict(tr, semantic=tr["semantic"]))
    rgb.update_batch(dict(position=tr["position"], yaw=tr["yaw"], elevation=tr["elevation"], depth=tr["depth"],
                          features=tr["rgb"]))
    for t in range(tr["depth"].shape[0]):
        base = dict(position=tr["position"][t], yaw=tr["yaw"][t], elevation=tr["elevation"][t], depth=tr["depth"][t])
        o_sem.update(dict(base, features=torch.nn.functional.one_hot(tr["semantic"][t], C).float()))
        o_rgb.update(dict(base, features=tr["rgb"][t]))
    return sem, rgb, o_sem, o_rgb


def test_episode_walkthrough_unshuffle_and_matching(device):
    from conftest import assert_map_close
    from mass_amd.utils.experimentation import predict_scene_differences, match_instances
    from oracle import massref as orc
    sem0, rgb0, o_sem0, o_rgb0 = build(device, scene(moved=False))
    sem1, rgb1, o_sem1, o_rgb1 = build(device, scene(moved=True, seed=0))
    for a, b in ((sem0, o_sem0), (rgb0, o_rgb0), (sem1, o_sem1), (rgb1, o_rgb1)):
        assert_map_close(a.data.cpu().numpy(), b.data.numpy())
    # the moved class is 5; class 3 did not move (distance below threshold)
    obj, goals0, goals1 = predict_scene_differences(sem0, sem1, rgb0, rgb1, set(), [3, 5], confidence_threshold=0.0,
                                                    contour_padding=0, distance_threshold=0.5)
    assert obj == 5 and len(goals0) == len(goals1) >= 1
    shift = (goals1[0] - goals0[0]).cpu().numpy()
    assert abs(shift[0] - 2.2) < 0.4 and abs(shift[1]) < 0.3          # moved ~2.2 m along x
    # oracle pipeline for the same class: find -> pairwise L2 -> scipy assignment
    w0 = orc.find(o_sem0.data, o_sem0.bins_x, o_sem0.bins_y, o_sem0.bins_z, 5, 0.0, 0, 0.0, o_rgb0.data)
    w1 = orc.find(o_sem1.data, o_sem1.bins_x, o_sem1.bins_y, o_sem1.bins_z, 5, 0.0, 0, 0.0, o_rgb1.data)
    c0, g0, s0, f0 = sem0.find(5, 0.0, 0, 0.0, rgb0)
    c1, g1, s1, f1 = sem1.find(5, 0.0, 0, 0.0, rgb1)
    assert sorted(sem0.boxes) == [d["box"] for d in w0] and sorted(sem1.boxes) == [d["box"] for d in w1]
    F0 = torch.stack([f0[sem0.boxes.index(d["box"])] for d in w0])
    F1 = torch.stack([f1[sem1.boxes.index(d["box"])] for d in w1])
    np.testing.assert_allclose(F0.cpu().numpy(), np.stack([d["feature"] for d in w0]), rtol=1e-4, atol=1e-6)
    cost, rows, cols = match_instances(F0, F1)
    want_cost, wr, wc = orc.match(np.stack([d["feature"] for d in w0]), np.stack([d["feature"] for d in w1]))
    np.testing.assert_allclose(cost.cpu().numpy(), want_cost, rtol=1e-3, atol=1e-5)
    assert np.array_equal(rows, wr) and np.array_equal(cols, wc)
    # the counters this episode would add to the final all-reduce
    from mass_amd.distributed import reduce_metrics
    m = reduce_metrics(dict(frames=20, n_matches=len(rows), map_abs_sum=float(sem0.data.abs().sum() + sem1.data.abs().sum())))
    assert m["frames"] == 20 and m["n_matches"] >= 1

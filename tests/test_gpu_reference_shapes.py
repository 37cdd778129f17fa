"""The shapes the reference's agent actually runs (agent.py:691-742, 825-832): 224 x 224 cameras, 384 x 384 x 96
maps at 0.05 m, C = 1 (occupancy), 54 (semantic) and - under --use-feature-matching - 256-d ResNet features
splatted at 56 x 56 with the depth sampled at the centre of every feature pixel
(resnet_projection_layer.py:120-131, 201-211).  384 * 384 * 96 * 256 = 3.6e9 floats: element offsets pass 2^31,
which the 64-bit offset paths of the dense-feature tile kernel, amax_z and roi_moments have to survive
("Index width").  Everything is compared with the oracle (slab by slab on the device for the
14.5 GB map)."""
import numpy as np
import pytest
import torch

from conftest import assert_map_close, assert_map_close_device

pytestmark = pytest.mark.gpu

SCREEN = 224
MAP = dict(map_height=384, map_width=384, map_depth=96, grid_resolution=0.05)


def trajectory(n, seed=0):
    from mass_amd.episodes import room_trajectory
    return room_trajectory(n, SCREEN, SCREEN, seed=seed)


@pytest.mark.parametrize("kind", ["semantic", "occupancy"])
def test_default_maps_per_frame_and_batched(device, kind):
    """Four per-frame update() calls, then one sequential batch of eight more frames, onto the agent's default map."""
    from oracle import massref as orc
    from mass_amd.nn.applications.occupancy_projection_layer import OccupancyProjectionLayer
    from mass_amd.nn.applications.semantic_projection_layer import SemanticProjectionLayer
    C = 54 if kind == "semantic" else 1
    kw = dict(camera_height=SCREEN, camera_width=SCREEN, vertical_fov=90.0, **MAP)
    lay = (SemanticProjectionLayer(feature_size=C, **kw) if kind == "semantic" else OccupancyProjectionLayer(**kw)).train().to(device)
    ref = orc.RefProjectionLayer(feature_size=C, **kw)
    tr = trajectory(12, seed=3)

    def feats(t):
        return torch.nn.functional.one_hot(tr["semantic"][t].long(), C).float() if kind == "semantic" else torch.ones(SCREEN, SCREEN, 1)
    for t in range(4):
        obs = dict(position=tr["position"][t], yaw=tr["yaw"][t], elevation=tr["elevation"][t], depth=tr["depth"][t])
        lay.update(dict(obs, semantic=tr["semantic"][t][..., None]) if kind == "semantic" else obs)
        ref.update(dict(obs, features=feats(t)))
    assert_map_close(lay.data.cpu().numpy(), ref.data.numpy(), what=f"{kind}: four updates")
    sl = slice(4, 12)
    batch = dict(position=tr["position"][sl], yaw=tr["yaw"][sl], elevation=tr["elevation"][sl], depth=tr["depth"][sl])
    if kind == "semantic":
        batch["semantic"] = tr["semantic"][sl]
    lay.update_batch(batch, sequential=True)
    for t in range(4, 12):
        ref.update(dict(position=tr["position"][t], yaw=tr["yaw"][t], elevation=tr["elevation"][t], depth=tr["depth"][t],
                        features=feats(t)))
    assert_map_close(lay.data.cpu().numpy(), ref.data.numpy(), what=f"{kind}: batch of eight")
    assert int((ref.data != 0).any(-1).sum()) > 10_000


def test_resnet_feature_map_beyond_2_to_31_elements(device):
    """256-d features at 56 x 56 onto 384 x 384 x 96 x 256 (14.5 GB): touched voxels lie on both sides of element
    offset 2^31; then amax_z over the whole map and find() with this map as the feature map (roi_moments)."""
    import psutil
    if psutil.virtual_memory().available < 48 * 2 ** 30:
        pytest.skip("the oracle's copy of the map needs 14.5 GB of host memory (plus slabs)")
    from oracle import massref as orc
    from mass_amd.nn.applications.resnet_projection_layer import ResNetProjectionLayer
    from mass_amd.nn.applications.semantic_projection_layer import SemanticProjectionLayer
    from mass_amd.utils.reductions import amax_z
    C = 256
    kw = dict(vertical_fov=90.0, **MAP)
    lay = ResNetProjectionLayer(camera_height=SCREEN, camera_width=SCREEN, feature_size=C, **kw).train()     # (no .cuda(): agent.py:725-742)
    ref = orc.RefProjectionLayer(camera_height=SCREEN // 4, camera_width=SCREEN // 4, feature_size=C, **kw)
    sem = SemanticProjectionLayer(camera_height=SCREEN, camera_width=SCREEN, feature_size=54, **kw).train().to(device)
    tr = trajectory(3, seed=9)
    g = torch.Generator().manual_seed(1)
    f = 4                                                                    # image_downsampling_factor (:201)
    for t in range(3):
        feats = torch.relu(torch.randn(SCREEN // 4, SCREEN // 4, C, generator=g))          # post-ReLU, like pseudo_forward's output
        obs = dict(position=tr["position"][t], yaw=tr["yaw"][t], elevation=tr["elevation"][t], depth=tr["depth"][t])
        lay.update(dict(obs, features=feats))
        sem.update(dict(obs, semantic=tr["semantic"][t][..., None]))
        ref.update(dict(obs, depth=tr["depth"][t][f // 2::f, f // 2::f], features=feats))
    assert lay.data.is_cuda and lay.data.numel() > 2 ** 31
    occupied = assert_map_close_device(lay.data, ref.data, what="256-d feature map")
    assert occupied > 1000
    # touched voxels on both sides of element offset 2^31 (row y of the map starts at y * 384 * 96 * 256 elements)
    rows = (ref.data != 0).any(-1).any(-1).any(-1).nonzero().reshape(-1)
    first_row_past = (2 ** 31) // (384 * 96 * C) + 1
    assert int(rows.min()) < first_row_past - 1 and int(rows.max()) > first_row_past, (int(rows.min()), int(rows.max()))
    # whole-map reduction with 64-bit offsets
    top = amax_z(lay.data)
    for y0 in range(0, 384, 96):
        want = lay.data[y0:y0 + 96].amax(dim=2)
        assert torch.equal(top[y0:y0 + 96], want), f"amax_z rows {y0}.."
    # find() on the semantic map with the 256-d map as feature map: expected features per detection
    labels, counts = torch.unique(tr["semantic"][0], return_counts=True)
    cls = int(labels[counts.argmax()])
    conf, coords, sizes, feats_found = sem.find(cls, confidence_threshold=0.0, contour_padding=0, feature_map=lay)
    assert len(conf) > 0 and len(feats_found) == len(conf)
    want = orc.find(sem.data.cpu(), sem.bins_x.cpu(), sem.bins_y.cpu(), sem.bins_z.cpu(), cls, confidence_threshold=0.0,
                    contour_padding=0, feature_data=ref.data)
    assert len(want) == len(conf)
    got_by_box = {tuple(int(v) for v in b): k for k, b in enumerate(sem.boxes)}
    for w in want:
        k = got_by_box[tuple(int(v) for v in w["box"])]
        np.testing.assert_allclose(float(conf[k]), w["confidence"], rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(feats_found[k].cpu().numpy().reshape(-1), np.asarray(w["feature"]).reshape(-1), rtol=2e-4, atol=1e-5)

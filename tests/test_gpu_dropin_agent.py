"""The reference agent's own construction of its maps, replayed through mass_amd.dropin.install(): the three
maps of a simulator step are built with ``.train().cuda()`` and the two ResNet feature maps of
``--use-feature-matching`` with ``.train()`` ONLY (agent.py:691-742).  Neither flag setting needs an edit of
the reference's scripts: a layer built on the CPU adopts the current HIP device at its first update
(BaseProjectionLayer._adopt_device).  One navigation_policy.py:164-171-style step and
predict_scene_differences (agent.py:435-444) run on them."""
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SCREEN, NUM_CLASSES = 224, 54          # SegmentationConfig.SCREEN_SIZE, NUM_CLASSES of the reference
MAP = dict(map_height=192, map_width=192, map_depth=64, grid_resolution=0.05)    # (the default 384 x 384 x 96 is covered by test_gpu_reference_shapes.py)


def test_agent_construction_runs_unchanged(device):
    import mass_amd.dropin
    mass_amd.dropin.install()
    try:
        from mass.nn.applications.occupancy_projection_layer import OccupancyProjectionLayer
        from mass.nn.applications.semantic_projection_layer import SemanticProjectionLayer
        from mass.nn.applications.resnet_projection_layer import ResNetProjectionLayer
        from mass_amd.utils.experimentation import predict_scene_differences
        from mass_amd.episodes import room_trajectory
        from oracle import massref as orc

        common = dict(camera_height=SCREEN, camera_width=SCREEN, vertical_fov=90.0, **MAP)
        # agent.py:691-718
        occupancy = OccupancyProjectionLayer(**common).train().cuda()
        semantic0 = SemanticProjectionLayer(feature_size=NUM_CLASSES, **common).train().cuda()
        semantic1 = SemanticProjectionLayer(feature_size=NUM_CLASSES, **common).train().cuda()
        # agent.py:723-742 (--use-feature-matching): no .cuda()
        resnet0 = ResNetProjectionLayer(feature_size=256, **common).train()
        resnet1 = ResNetProjectionLayer(feature_size=256, **common).train()
        assert resnet0.data.device.type == "cpu" and occupancy.data.is_cuda

        tr = room_trajectory(3, SCREEN, SCREEN, seed=5)
        ref = orc.RefProjectionLayer(feature_size=NUM_CLASSES, **common)
        for t in range(3):
            obs = dict(position=tr["position"][t].numpy(), yaw=float(tr["yaw"][t]), elevation=float(tr["elevation"][t]),
                       depth=tr["depth"][t].numpy(), semantic=tr["semantic"][t].numpy().astype(np.int64)[..., None],
                       rgb=tr["rgb"][t].numpy())
            # navigation_policy.py:164-171: every map takes the observation in turn
            for layer in (occupancy, semantic0, resnet0):
                layer.update(obs)
            for layer in (semantic1, resnet1):
                layer.update(obs)
            ref.update(dict(position=tr["position"][t], yaw=tr["yaw"][t], elevation=tr["elevation"][t], depth=tr["depth"][t],
                            features=torch.nn.functional.one_hot(tr["semantic"][t].long(), NUM_CLASSES).float()))
        assert resnet0.data.is_cuda and resnet1.data.is_cuda, "a CPU-built layer adopts the HIP device at its first update"
        assert tuple(resnet0.data.shape) == (192, 192, 64, 256) and float(resnet0.data.abs().sum()) > 0
        from conftest import assert_map_close
        assert_map_close(semantic0.data.cpu().numpy(), ref.data.numpy(), what="semantic map of the replayed step")
        # agent.py:435-444
        out = predict_scene_differences(semantic0, semantic1, resnet0, resnet1, set(), list(range(NUM_CLASSES)),
                                        confidence_threshold=0.0, contour_padding=0, distance_threshold=0.5)
        assert len(out) == 3
    finally:
        for m in [k for k in sys.modules if k.startswith(("mass.", "slam_rcnn")) or k in ("mass", "slam_rcnn")]:
            del sys.modules[m]

"""the ResNet-50 stem + layer1 front end of ResNetProjectionLayer
(/root/reference/mass/nn/applications/resnet_projection_layer.py:143-157).  torchvision and its
weights are absent here, so the check is structural: the published architecture's parameter names
and counts, output shape [H/4, W/4, 256], stride 4, post-ReLU, determinism; parity unpinned."""
import numpy as np
import pytest
import torch


def test_architecture_matches_the_published_resnet50_prefix():
    from mass_amd.nn.models.resnet_stem import ResNet50Layer1
    m = ResNet50Layer1()
    assert sum(p.numel() for p in m.parameters()) == 225_344          # conv1 + bn1 + layer1 of resnet50
    keys = set(m.state_dict())
    for k in ("conv1.weight", "bn1.running_mean", "layer1.0.conv1.weight", "layer1.0.downsample.0.weight",
              "layer1.0.downsample.1.running_var", "layer1.2.conv3.weight", "layer1.2.bn3.bias"):
        assert k in keys
    assert tuple(m.conv1.weight.shape) == (64, 3, 7, 7) and m.conv1.stride == (2, 2)
    assert tuple(m.layer1[0].conv2.weight.shape) == (64, 64, 3, 3)
    assert tuple(m.layer1[2].conv3.weight.shape) == (256, 64, 1, 1)
    # a full torchvision-style state dict loads (extra keys ignored), an incomplete one is refused
    full = dict(m.state_dict(), **{"fc.weight": torch.zeros(1000, 2048), "layer2.0.conv1.weight": torch.zeros(128, 256, 1, 1)})
    ResNet50Layer1().load_torchvision_state_dict(full)
    with pytest.raises(KeyError):
        ResNet50Layer1().load_torchvision_state_dict({"conv1.weight": m.conv1.weight})


def test_forward_shape_stride_relu_and_preprocess():
    from mass_amd.nn.models.resnet_stem import ResNet50Layer1, ResNetFeatureExtractor, preprocess
    g = torch.Generator().manual_seed(0)
    rgb = torch.rand(224, 224, 3, generator=g)
    x = preprocess(rgb)
    assert tuple(x.shape) == (1, 3, 224, 224)
    q = np.uint8(255.0 * rgb.numpy()).astype(np.float32) / 255.0      # the reference quantises to uint8 first
    np.testing.assert_allclose(x[0, 1].numpy(), (q[..., 1] - 0.456) / 0.224, rtol=1e-6, atol=1e-6)
    assert tuple(preprocess(torch.rand(120, 160, 3, generator=g)).shape) == (1, 3, 224, 298)    # shorter side -> 224
    ex = ResNetFeatureExtractor(torch.device("cpu"))
    f = ex(rgb)
    assert tuple(f.shape) == (56, 56, 256) and f.dtype == torch.float32
    assert float(f.min()) >= 0.0 and float(f.max()) > 0.0            # post-ReLU
    assert torch.equal(f, ResNetFeatureExtractor(torch.device("cpu"))(rgb))   # seeded initialisation
    # stride 4: an input shifted by 4 pixels shifts the interior of the feature image by 1
    m = ex.model
    # (receptive field of layer1: 35 input pixels, so a margin of 6 feature pixels is kept)
    xa = torch.randn(1, 3, 128, 128, generator=g)
    xb = torch.roll(xa, shifts=4, dims=3)
    fa, fb = m(xa), m(xb)
    torch.testing.assert_close(fa[..., 6:-6, 6:-7], fb[..., 6:-6, 7:-6], rtol=1e-4, atol=1e-5)
    assert not torch.allclose(fa[..., 6:-6, 6:-7], fb[..., 6:-6, 6:-7], atol=1e-3)


@pytest.mark.gpu
def test_resnet_layer_update_on_device(device):
    """End to end on the MI355X: RGB frame -> stem + layer1 (MIOpen) -> 256-d splat (HIP), against the
    oracle fed with the same features; the CPU and GPU feature images agree."""
    from oracle import massref as orc
    from conftest import assert_map_close
    from mass_amd.nn.applications.resnet_projection_layer import ResNetProjectionLayer
    from mass_amd.nn.models.resnet_stem import ResNetFeatureExtractor
    g = torch.Generator().manual_seed(3)
    H = W = 224
    kw = dict(map_height=48, map_width=48, map_depth=24, grid_resolution=0.1)
    lay = ResNetProjectionLayer(camera_height=H, camera_width=W, feature_size=256, **kw).to(device)
    ref = orc.RefProjectionLayer(camera_height=H // 4, camera_width=W // 4, feature_size=256, **kw)
    cpu_ex = ResNetFeatureExtractor(torch.device("cpu"))
    for t in range(2):
        rgb = torch.rand(H, W, 3, generator=g)
        depth = 0.5 + 1.5 * torch.rand(H, W, 1, generator=g)
        obs = dict(position=np.asarray((0.05 * t, 0.0, 0.1), np.float32), yaw=0.4 + 0.3 * t, elevation=-0.3, depth=depth)
        lay.update(dict(obs, rgb=rgb))
        feats = lay.feature_extractor(rgb)
        assert feats.is_cuda and tuple(feats.shape) == (56, 56, 256)
        torch.testing.assert_close(feats.cpu(), cpu_ex(rgb), rtol=2e-3, atol=2e-3)
        ref.update(dict(obs, depth=depth[2::4, 2::4], features=feats.cpu()))
    assert_map_close(lay.data.cpu().numpy(), ref.data.numpy(), what="resnet feature map")
    out = lay.pseudo_forward(torch.randn(1, 3, 64, 64))
    assert tuple(out.shape) == (1, 256, 16, 16)

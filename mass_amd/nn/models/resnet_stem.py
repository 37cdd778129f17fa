"""ResNet-50 stem + layer1 — the front end of the reference's ResNetProjectionLayer.

The reference runs ``conv1 -> bn1 -> relu -> maxpool -> layer1`` of torchvision's resnet50 on the
RGB frame (/root/reference/mass/nn/applications/resnet_projection_layer.py:143-157, ``pseudo_forward``)
and splats the resulting 256-channel, stride-4, post-ReLU feature image.  torchvision is not part
of this image and the pretrained weights are a remote download (:134), so the architecture is
written out here from its published definition (He et al. 2016, bottleneck blocks with the stride
on the 3x3 convolution; 225,344 parameters up to layer1) with torchvision's parameter names:
a locally available resnet50 ``state_dict`` loads with ``load_torchvision_state_dict``.  The
convolutions run through torch (MIOpen on the MI355X) — SURVEY 8(f4): not hand-written HIP.
"""
from typing import Optional

import numpy as np
import torch
from torch import nn

IMAGENET_MEAN = (0.485, 0.456, 0.406)      # resnet_projection_layer.py:139-140
IMAGENET_STD = (0.229, 0.224, 0.225)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes: int, planes: int, downsample: Optional[nn.Module] = None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * self.expansion, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        identity = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.relu(out + identity)


class ResNet50Layer1(nn.Module):
    """conv1 (7x7/2) - bn1 - relu - maxpool (3x3/2) - layer1 (3 bottlenecks, 64 -> 256 channels)."""

    out_channels = 256
    stride = 4

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        downsample = nn.Sequential(nn.Conv2d(64, 256, kernel_size=1, stride=1, bias=False), nn.BatchNorm2d(256))
        self.layer1 = nn.Sequential(Bottleneck(64, 64, downsample), Bottleneck(256, 64), Bottleneck(256, 64))
        for m in self.modules():                       # torchvision's initialisation
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    @torch.no_grad()
    def forward(self, x):
        """[N, 3, H, W] normalised RGB -> [N, 256, ceil(H/4), ceil(W/4)], post-ReLU (pseudo_forward, :143-157)."""
        return self.layer1(self.maxpool(self.relu(self.bn1(self.conv1(x)))))

    def load_torchvision_state_dict(self, state_dict):
        """Take conv1 / bn1 / layer1 from a full torchvision resnet50 state dict (the rest is ignored)."""
        own = self.state_dict()
        picked = {k: v for k, v in state_dict.items() if k in own}
        missing = sorted(set(own) - set(picked))
        if missing:
            raise KeyError(f"state dict lacks {missing[:4]}{'...' if len(missing) > 4 else ''}")
        self.load_state_dict(picked)
        return self


def preprocess(rgb, size: int = 224):
    """The reference's transform (:135-141) for an [H, W, 3] image in [0, 1]: quantise to uint8, resize the
    shorter side to `size` (PIL bilinear, as torchvision's Resize does on a PIL image), scale to [0, 1],
    normalise with the ImageNet statistics.  Returns a CPU tensor [1, 3, h, w]."""
    arr = np.uint8(255.0 * np.asarray(torch.as_tensor(rgb).detach().cpu().numpy() if isinstance(rgb, torch.Tensor) else rgb))
    h, w = arr.shape[:2]
    if min(h, w) != size:
        from PIL import Image
        if h <= w:
            nh, nw = size, int(size * w / h)
        else:
            nh, nw = int(size * h / w), size
        arr = np.asarray(Image.fromarray(arr).convert("RGB").resize((nw, nh), Image.BILINEAR))
    x = torch.from_numpy(np.array(arr, dtype=np.uint8)).permute(2, 0, 1).to(torch.float32).div_(255.0)
    mean = torch.tensor(IMAGENET_MEAN).view(3, 1, 1)
    std = torch.tensor(IMAGENET_STD).view(3, 1, 1)
    return ((x - mean) / std).unsqueeze(0)


class ResNetFeatureExtractor:
    """rgb [H, W, 3] in [0, 1] -> features [h, w, 256] on `device`, what the reference computes per
    frame before the splat (:197-199).  `weights`: path of a torchvision resnet50 checkpoint to load
    (safe tensors-only load); without it the network keeps its random initialisation."""

    def __init__(self, device, weights: Optional[str] = None, seed: int = 0):
        with torch.random.fork_rng():
            torch.manual_seed(seed)
            self.model = ResNet50Layer1()
        if weights is not None:
            self.model.load_torchvision_state_dict(torch.load(weights, map_location="cpu", weights_only=True))
        self.model = self.model.eval().to(device)
        self.device = device

    def __call__(self, rgb):
        x = preprocess(rgb).to(self.device)
        return self.model(x).squeeze(0).permute(1, 2, 0).contiguous()

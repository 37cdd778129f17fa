"""ResNetProjectionLayer — dense image features splatted at 1/4 camera resolution.

Mirrors /root/reference/mass/nn/applications/resnet_projection_layer.py for the part that is
on the hot path: the layer is built at camera_height // 4 x camera_width // 4 (:120-131) and
``update`` splats a [h, w, C] fp32 feature image with the depth sampled at the centre of each
feature pixel, ``depth[f // 2::f, f // 2::f]`` (:201-211), through the HIP pipeline.

The feature extractor itself (ResNet-50 stem + layer1 on the RGB frame, :143-157) is out of
scope (its weights are a remote download, and convolution belongs to MIOpen):
pass any callable ``feature_extractor(rgb [H, W, 3] in [0, 1]) -> [h, w, C]`` tensor.  Without
one the layer tries torchvision's resnet50 with locally available weights and fails loudly if
that is not possible.
"""
from typing import Callable, Dict, Optional

import numpy as np
import torch

from mass_amd.nn.base_projection_layer import BaseProjectionLayer


def _torchvision_layer1(device):
    try:
        from torchvision.models import resnet50
    except Exception as exc:                      # torchvision is not in this image
        raise ImportError("ResNetProjectionLayer needs a feature_extractor (torchvision is not "
                          "installed, and the reference's pretrained weights are a remote download)") from exc
    model = resnet50(weights=None).eval().to(device)
    mean = torch.tensor([0.485, 0.456, 0.406], device=device).view(1, 3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225], device=device).view(1, 3, 1, 1)

    def extract(rgb):
        x = torch.as_tensor(rgb, dtype=torch.float32, device=device).permute(2, 0, 1).unsqueeze(0)
        scale = 224.0 / min(x.shape[-2:])
        x = torch.nn.functional.interpolate(x, scale_factor=scale, mode="bilinear", antialias=True)
        x = (x - mean) / std
        with torch.no_grad():
            x = model.layer1(model.maxpool(model.relu(model.bn1(model.conv1(x)))))
        return x.squeeze(0).permute(1, 2, 0)
    return extract


class ResNetProjectionLayer(BaseProjectionLayer):

    def __init__(self, camera_height: int = 224, camera_width: int = 224,
                 vertical_fov: float = 90.0, map_height: int = 256,
                 map_width: int = 256, map_depth: int = 64,
                 feature_size: int = 256, dtype: torch.dtype = torch.float32,
                 origin_y: float = 0.0, origin_x: float = 0.0,
                 origin_z: float = 0.0, grid_resolution: float = 0.05,
                 interpolation_weight: float = 0.5,
                 initial_feature_map: torch.Tensor = None,
                 feature_extractor: Optional[Callable] = None):
        super(ResNetProjectionLayer, self).__init__(
            camera_height=camera_height // 4, camera_width=camera_width // 4,
            vertical_fov=vertical_fov, map_height=map_height,
            map_width=map_width, map_depth=map_depth,
            feature_size=feature_size, dtype=dtype,
            origin_y=origin_y, origin_x=origin_x, origin_z=origin_z,
            grid_resolution=grid_resolution,
            interpolation_weight=interpolation_weight,
            initial_feature_map=initial_feature_map)
        self.feature_extractor = feature_extractor

    def update(self, observation: Dict[str, torch.Tensor]):
        """resnet_projection_layer.py:159-213.  Keys: position, yaw, elevation, depth [H, W, 1],
        rgb [H, W, 3] in [0, 1] (or, as an extension, precomputed ``features`` [h, w, C])."""
        depth = torch.as_tensor(observation["depth"], dtype=torch.float32, device=self.data.device)
        if "features" in observation:
            features = torch.as_tensor(observation["features"], dtype=torch.float32, device=self.data.device)
        else:
            if self.feature_extractor is None:
                self.feature_extractor = _torchvision_layer1(self.data.device)
            features = torch.as_tensor(self.feature_extractor(observation["rgb"]), dtype=torch.float32,
                                       device=self.data.device)
        f = depth.shape[0] // features.shape[0]               # image_downsampling_factor (:201)
        super(ResNetProjectionLayer, self).update(
            dict(position=observation["position"], yaw=observation["yaw"], elevation=observation["elevation"],
                 depth=depth[f // 2::f, f // 2::f], features=features))
        return self

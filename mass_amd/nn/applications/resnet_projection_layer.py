"""ResNetProjectionLayer — dense image features splatted at 1/4 camera resolution.

Mirrors /root/reference/mass/nn/applications/resnet_projection_layer.py: the layer is built at
camera_height // 4 x camera_width // 4 (:120-131); ``update`` runs the ResNet-50 stem + layer1 on
the RGB frame (``pseudo_forward``, :143-157; here mass_amd.nn.models.resnet_stem, torch / MIOpen)
and splats the [h, w, 256] fp32 feature image with the depth sampled at the centre of each
feature pixel, ``depth[f // 2::f, f // 2::f]`` (:201-211), through the HIP pipeline.

The reference downloads pretrained weights (``resnet50(pretrained=True)``, :134): offline that is
impossible, so the default extractor keeps a seeded random initialisation unless ``weights`` names
a local torchvision resnet50 checkpoint.  Any callable ``feature_extractor(rgb) -> [h, w, C]``
can be plugged in instead.
"""
from typing import Callable, Dict, Optional

import numpy as np
import torch

from mass_amd.nn.base_projection_layer import BaseProjectionLayer
from mass_amd.nn.models.resnet_stem import ResNetFeatureExtractor


class ResNetProjectionLayer(BaseProjectionLayer):

    def __init__(self, camera_height: int = 224, camera_width: int = 224,
                 vertical_fov: float = 90.0, map_height: int = 256,
                 map_width: int = 256, map_depth: int = 64,
                 feature_size: int = 256, dtype: torch.dtype = torch.float32,
                 origin_y: float = 0.0, origin_x: float = 0.0,
                 origin_z: float = 0.0, grid_resolution: float = 0.05,
                 interpolation_weight: float = 0.5,
                 initial_feature_map: torch.Tensor = None,
                 feature_extractor: Optional[Callable] = None,
                 weights: Optional[str] = None):
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
        self.weights = weights

    def pseudo_forward(self, x):
        """resnet_projection_layer.py:143-157 on an already normalised [N, 3, H, W] batch."""
        if self.feature_extractor is None:
            self.feature_extractor = ResNetFeatureExtractor(self.data.device, weights=self.weights)
        return self.feature_extractor.model(x.to(self.data.device))

    def update(self, observation: Dict[str, torch.Tensor]):
        """resnet_projection_layer.py:159-213.  Keys: position, yaw, elevation, depth [H, W, 1],
        rgb [H, W, 3] in [0, 1] (or, as an extension, precomputed ``features`` [h, w, C])."""
        self._adopt_device()            # (built without .cuda(), like agent.py:721-742 builds these layers)
        depth = torch.as_tensor(observation["depth"], dtype=torch.float32, device=self.data.device)
        if "features" in observation:
            features = torch.as_tensor(observation["features"], dtype=torch.float32, device=self.data.device)
        else:
            if self.feature_extractor is None:
                self.feature_extractor = ResNetFeatureExtractor(self.data.device, weights=self.weights)
            features = torch.as_tensor(self.feature_extractor(observation["rgb"]), dtype=torch.float32,
                                       device=self.data.device)
        f = depth.shape[0] // features.shape[0]               # image_downsampling_factor (:201)
        super(ResNetProjectionLayer, self).update(
            dict(position=observation["position"], yaw=observation["yaw"], elevation=observation["elevation"],
                 depth=depth[f // 2::f, f // 2::f], features=features))
        return self

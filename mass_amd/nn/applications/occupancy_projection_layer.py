"""OccupancyProjectionLayer — C = 1 map of ray end points.

Mirrors /root/reference/mass/nn/applications/occupancy_projection_layer.py:
``update`` is the base update with features = ones_like(depth) (:159-161),
which the HIP pipeline evaluates as MF_FEAT_ONES without a feature tensor.
"""
from typing import Any, Dict

import torch

from mass_amd.nn.base_projection_layer import BaseProjectionLayer


class OccupancyProjectionLayer(BaseProjectionLayer):

    def __init__(self, camera_height: int = 224, camera_width: int = 224,
                 vertical_fov: float = 90.0, map_height: int = 256,
                 map_width: int = 256, map_depth: int = 64,
                 dtype: torch.dtype = torch.float32,
                 origin_y: float = 0.0, origin_x: float = 0.0,
                 origin_z: float = 0.0, grid_resolution: float = 0.05,
                 interpolation_weight: float = 0.5,
                 initial_feature_map: torch.Tensor = None):
        super(OccupancyProjectionLayer, self).__init__(
            camera_height=camera_height, camera_width=camera_width,
            vertical_fov=vertical_fov, map_height=map_height,
            map_width=map_width, map_depth=map_depth,
            feature_size=1, dtype=dtype,
            origin_y=origin_y, origin_x=origin_x,
            origin_z=origin_z, grid_resolution=grid_resolution,
            interpolation_weight=interpolation_weight,
            initial_feature_map=initial_feature_map)

    def update(self, observation: Dict[str, torch.Tensor]):
        """Mark the voxels around every ray end point as occupied
        (occupancy_projection_layer.py:122-163)."""
        self._splat(observation["position"], observation["yaw"], observation["elevation"],
                    observation["depth"], None)
        return self

    def update_batch(self, observation: Dict[str, torch.Tensor], sequential: bool = True):
        self._splat(observation["position"], observation["yaw"], observation["elevation"],
                    observation["depth"], None, sequential=sequential)
        return self

    def visualize(self, obs: Dict[str, Any], depth_slice: slice = slice(4, 32)):
        # the reference draws the agent path with cv2 (mass/utils/visualization.py,
        # out of scope: debug video); the free-space image itself is the base one
        return super(OccupancyProjectionLayer, self).visualize(obs, depth_slice=depth_slice)

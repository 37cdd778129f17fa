"""SemanticProjectionLayer — C-class semantic voxel map.

Mirrors /root/reference/mass/nn/applications/semantic_projection_layer.py:
``update`` takes ``observation["semantic"]`` [H, W, 1] class ids (:203-209).
The reference expands them to a one-hot fp32 [H, W, C] image (66 MB at
480x640x54) and splats that; here the ids go to the HIP pipeline as
MF_FEAT_LABEL_* and the one-hot tensor is never materialised.
"""
from typing import Any, Dict

import numpy as np
import torch
import torch.nn.functional as functional

from mass_amd import _lib
from mass_amd._lib import lib, check, ptr, current_stream, MAX_FRAMES_PER_CALL
from mass_amd.nn.base_projection_layer import BaseProjectionLayer
from mass_amd.utils.reductions import amax_z


def contour_boxes(image, reverse_order=True):
    """Bounding boxes (x, y, w, h) of all borders of a binary image, in the order
    cv2.findContours(RETR_LIST) lists them (host; mf_contour_boxes)."""
    image = np.ascontiguousarray(image, dtype=np.uint8)
    h, w = image.shape
    cap = max(16, (h * w + 1) // 2 + 8)
    boxes = np.empty((cap, 4), np.int32)
    n = check(lib.mf_contour_boxes(image.ctypes.data, h, w, int(reverse_order), boxes.ctypes.data, cap))
    return boxes[:min(n, cap)]


class SemanticProjectionLayer(BaseProjectionLayer):

    def __init__(self, camera_height: int = 224, camera_width: int = 224,
                 vertical_fov: float = 90.0, map_height: int = 256,
                 map_width: int = 256, map_depth: int = 64,
                 feature_size: int = 1, dtype: torch.dtype = torch.float32,
                 origin_y: float = 0.0, origin_x: float = 0.0,
                 origin_z: float = 0.0, grid_resolution: float = 0.05,
                 interpolation_weight: float = 0.5,
                 initial_feature_map: torch.Tensor = None,
                 class_to_colors: torch.Tensor = None):
        super(SemanticProjectionLayer, self).__init__(
            camera_height=camera_height, camera_width=camera_width,
            vertical_fov=vertical_fov, map_height=map_height,
            map_width=map_width, map_depth=map_depth,
            feature_size=feature_size, dtype=dtype,
            origin_y=origin_y, origin_x=origin_x,
            origin_z=origin_z, grid_resolution=grid_resolution,
            interpolation_weight=interpolation_weight,
            initial_feature_map=initial_feature_map)
        self.boxes = None
        if class_to_colors is not None:
            self.register_buffer('class_to_colors', class_to_colors)
        else:
            self.class_to_colors = None

    def reset(self, origin_y: float = 0.0, origin_x: float = 0.0, origin_z: float = 0.0):
        self.check_labels(synchronize=False)
        self.boxes = None
        super(SemanticProjectionLayer, self).reset(
            origin_y=origin_y, origin_x=origin_x, origin_z=origin_z)

    def _labels(self, semantic):
        """Class-id image for the kernel: [H, W] (or [B, H, W]) uint8 / int32 / int64 on the
        device.  Host arrays are uploaded as they are (an int64 480x640 image is 2.4 MB and takes
        ~60 us; narrowing it on the host first costs far more than it saves)."""
        semantic = torch.as_tensor(semantic)
        if semantic.is_floating_point():
            semantic = semantic.to(torch.int64)
        if semantic.dim() >= 3 and semantic.shape[-1] == 1:
            semantic = semantic[..., 0]
        semantic = semantic.to(device=self.data.device)
        if semantic.dtype not in (torch.uint8, torch.int32, torch.int64):
            semantic = semantic.to(torch.int64)
        return semantic

    _CLASS_ERROR = "Class values must be non-negative and smaller than num_classes."

    def _status(self):
        """Pinned host word the kernels raise when a class id is outside [0, C)."""
        if getattr(self, "_label_status", None) is None:
            self._label_status = torch.zeros(1, dtype=torch.int32).pin_memory()
        return self._label_status

    def check_labels(self, synchronize: bool = True):
        """Raise if an update since the last check met a class id outside [0, feature_size)
        (such an update left the map untouched).  validate="defer" updates rely on this."""
        st = getattr(self, "_label_status", None)
        if st is None:
            return
        if synchronize:
            torch.cuda.current_stream(self.data.device).synchronize()
        if int(st[0]) != 0:
            st[0] = 0
            raise RuntimeError(self._CLASS_ERROR)

    def _update(self, observation, sequential, validate):
        self._adopt_device()            # (a layer built without .cuda())
        if validate:
            self.check_labels(synchronize=False)         # whatever an earlier (deferred) update has reported by now
        labels = self._labels(observation["semantic"])
        if validate is True and sequential and labels.dim() == 3 and labels.shape[0] > MAX_FRAMES_PER_CALL:
            # such a batch is issued in several library calls: look at all of its ids first, so that a raise
            # leaves the map untouched like a single call does
            if bool(((labels < 0) | (labels >= self.feature_size)).any()):
                raise RuntimeError(self._CLASS_ERROR)
        self._splat(observation["position"], observation["yaw"], observation["elevation"],
                    observation["depth"], labels, sequential=sequential,
                    label_status=self._status() if validate else None)
        if validate is True:
            self.check_labels(synchronize=True)
        return self

    def update(self, observation: Dict[str, torch.Tensor], validate=True):
        """semantic_projection_layer.py:165-216.  Class ids outside [0, feature_size) raise like the
        reference's one_hot (:203-209) and leave the map untouched: the kernels detect them and call
        the update off.  validate=True (default, the reference's behaviour) waits for this update's
        kernels (one stream synchronisation) and raises from this very call; validate="defer" does not
        wait: the RuntimeError surfaces at the next update() / find() / reset() / check_labels() of this
        layer, the way device-side errors usually do (callers that read .data directly call
        check_labels() first); False skips the check (such ids then count as an all-zero feature row).
        INTEGRATION.md has the table."""
        return self._update(observation, True, validate)

    def update_batch(self, observation: Dict[str, torch.Tensor], sequential: bool = True, validate=True):
        return self._update(observation, sequential, validate)

    # ------------------------------------------------------------------ find
    def _voxel_centres(self):
        """World coordinate of the voxel centre per index (what map_to_world returns for
        integer coordinates, base_projection_layer.py:479-511); y is stored flipped."""
        cx = (self.bins_x[:-1] + self.bins_x[1:]) / 2
        cy = ((self.bins_y[:-1] + self.bins_y[1:]) / 2).flip(-1)
        cz = (self.bins_z[:-1] + self.bins_z[1:]) / 2
        return cx.contiguous(), cy.contiguous(), cz.contiguous()

    def class_images(self, contour_threshold: float = 0.0):
        """[H, W, C] bool: (data > threshold).any(dim=2) for every class in ONE pass over the
        map (the reference re-reads the whole map once per class).  Cached until the map
        changes through update()/reset()."""
        key = (self._map_version, float(contour_threshold))
        if getattr(self, "_class_images", None) is None or self._class_images[0] != key:
            self._class_images = (key, amax_z(self.data) > contour_threshold)
        return self._class_images[1]

    def find(self, semantic_category: int, confidence_threshold: float = 0.2,
             contour_padding: int = 3, contour_threshold: float = 0.0,
             feature_map=None):
        """Instances of one class: per connected blob of the top-down class image, the expected
        world position, confidence, size in voxels and (optionally) the expected feature vector
        (semantic_projection_layer.py:257-362).  Returns (confidences, coordinates, sizes,
        features or None) as lists of tensors, one entry per detection."""
        self.check_labels(synchronize=False)
        data = self.data
        c = int(semantic_category)
        if contour_padding == 0:
            image = self.class_images(contour_threshold)[..., c]
        else:
            # smoothing with a (2p+1)^3 box filter first (avg_pool3d, zero padded): plain torch
            smooth = functional.avg_pool3d(data[..., c][None, None], contour_padding * 2 + 1, stride=1,
                                           padding=contour_padding)[0, 0]
            image = (smooth > contour_threshold).any(dim=2)
        boxes = contour_boxes(image.to(torch.uint8).cpu().numpy())

        self.boxes, coordinates, confidences, sizes, features = [], [], [], [], []
        want_feat = feature_map is not None
        if len(boxes) == 0:
            return confidences, coordinates, sizes, features if want_feat else None

        cx, cy, cz = self._voxel_centres()
        dev = data.device
        boxes_d = torch.as_tensor(boxes, device=dev)
        n = len(boxes)
        moments = torch.empty(n, 5, dtype=torch.float32, device=dev)
        fdata = feat_out = None
        on_device = want_feat and feature_map.data.device == dev
        if on_device:
            fdata = feature_map.data
            if fdata.dtype != torch.float32 or not fdata.is_contiguous() or fdata.shape[:3] != data.shape[:3]:
                raise ValueError("feature_map.data must be a contiguous float32 map of the same grid")
            feat_out = torch.empty(n, fdata.shape[-1], dtype=torch.float32, device=dev)
        check(lib.mf_roi_moments(ptr(data), data.shape[0], data.shape[1], data.shape[2], data.shape[3], c,
                                 ptr(cx), ptr(cy), ptr(cz), ptr(boxes_d), n, ptr(fdata),
                                 fdata.shape[-1] if on_device else 0, ptr(moments), ptr(feat_out),
                                 current_stream(dev)))
        s1, s2 = moments[:, 0], moments[:, 1]
        den = s1 + 1e-9                                   # weights = mask_roi / (mask_roi.sum() + 1e-9)
        conf = s2 / den
        centre = moments[:, 2:5] / den[:, None]
        keep = (conf > confidence_threshold).cpu().numpy()
        for k in range(n):
            if not keep[k]:
                continue
            x, y, w, h = (int(v) for v in boxes[k])
            self.boxes.append((x, y, w, h))
            confidences.append(conf[k])
            coordinates.append(centre[k])
            sizes.append(s1[k])
            if want_feat:
                if on_device:
                    features.append(feat_out[k] / den[k])
                else:                                     # e.g. a feature map kept on the CPU (agent.py:711-742)
                    roi = feature_map.data[y:y + h, x:x + w].to(dev)
                    wts = data[y:y + h, x:x + w, :, c:c + 1] / den[k]
                    features.append((roi * wts).sum(dim=(0, 1, 2)))
        return confidences, coordinates, sizes, features if want_feat else None

    def visualize(self, obs: Dict[str, Any], depth_slice: slice = slice(0, 32)):
        """Top-down class colour image (semantic_projection_layer.py:218-255,
        without the cv2 box overlay)."""
        if self.class_to_colors is None:
            raise ValueError("class_to_colors was not given")
        top_down_map = self.top_down(depth_slice=depth_slice)
        image = torch.nn.functional.embedding(top_down_map.argmax(dim=-1), self.class_to_colors)
        image = torch.where((top_down_map != 0).any(dim=-1, keepdim=True),
                            image, torch.ones_like(image)).cpu().numpy()
        return image

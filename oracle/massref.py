"""oracle/massref.py — Python face of the CPU oracle (oracle/massref.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never by the product package ``mass_amd``.

The functions keep the reference's names, argument order and return
conventions so parity tests read like calls into the reference:

    spherical_to_cartesian  /root/reference/mass/utils/projection.py:6-31
    project_camera_rays     projection.py:34-74
    transform_rays          projection.py:77-110
    bin_rays                projection.py:113-230
    update_feature_map      projection.py:233-351
    RefProjectionLayer      mass/nn/base_projection_layer.py:67-181,183-235,282-343
    pairwise_l2, match      mass/utils/experimentation.py:261-287

Pose trigonometry and the rotation matrix are O(1) and are evaluated with the
same torch ops the reference uses (so libm never enters the comparison); all
per-pixel and per-voxel arithmetic runs in the C restatement.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmassref.so")


def build(force=False):
    """Compile oracle/massref.c -> oracle/libmassref.so (gcc, see Makefile)."""
    src = os.path.join(_HERE, "massref.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libmassref.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def _default_threads():
    """Threads of the blend loop (bit-identical results for any count, see massref.c): MASSREF_THREADS, else
    the CPUs this process may use, at most 8 (more threads than a container's CPU share make it slower: every
    thread walks all contributions and only the per-voxel work is divided)."""
    env = os.environ.get("MASSREF_THREADS")
    if env:
        return max(1, int(env))
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(n, 8))


def set_threads(n):
    lib().ref_set_threads(int(n))


def get_threads():
    return lib().ref_get_threads()


def force_threads(on):
    """Run the blend loop on the set number of threads whatever the scene (by default it takes one thread when the
    voxels receive few contributions each: such frames are memory bound and threads make them slower)."""
    lib().ref_force_threads(1 if on else 0)


def last_threads():
    return lib().ref_last_threads()


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        i64, f32p, i64p, u8p = ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p
        L.ref_project_camera_rays.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                              ctypes.c_double, f32p]
        L.ref_project_camera_rays.restype = None
        L.ref_transform_rays.argtypes = [i64, f32p, f32p, f32p]
        L.ref_transform_rays.restype = None
        L.ref_bin_rays.argtypes = [f32p, ctypes.c_int, f32p, ctypes.c_int, f32p, ctypes.c_int,
                                   i64, i64, f32p, f32p, f32p, ctypes.c_float, ctypes.c_float,
                                   i64p, i64p, i64p, f32p, f32p, f32p, f32p, u8p]
        L.ref_bin_rays.restype = None
        L.ref_update_feature_map.argtypes = [i64, i64p, i64p, i64p, f32p, f32p, f32p, f32p, f32p,
                                             i64, i64, i64, i64, ctypes.c_float, i64p, i64]
        L.ref_update_feature_map.restype = i64
        L.ref_pairwise_l2.argtypes = [f32p, i64, f32p, i64, i64, f32p]
        L.ref_pairwise_l2.restype = None
        L.ref_set_threads.argtypes = [ctypes.c_int]
        L.ref_set_threads.restype = None
        L.ref_get_threads.restype = ctypes.c_int
        L.ref_force_threads.argtypes = [ctypes.c_int]
        L.ref_force_threads.restype = None
        L.ref_last_threads.restype = ctypes.c_int
        L.ref_set_threads(_default_threads())
        _lib = L
    return _lib


def _f32(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float32))


def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data) if a is not None else None


# --------------------------------------------------------------------------
# functional API (same call surface as mass.utils.projection)
# --------------------------------------------------------------------------

def spherical_to_cartesian(yaw, elevation):
    """projection.py:29-31 (same torch ops, kept on the host)."""
    yaw = torch.as_tensor(yaw, dtype=torch.float32)
    elevation = torch.as_tensor(elevation, dtype=torch.float32)
    return torch.stack([torch.cos(yaw) * torch.cos(elevation),
                        torch.sin(yaw) * torch.cos(elevation),
                        torch.sin(elevation)], dim=-1)


def rotation_from(eye_vector, up_vector):
    """projection.py:104-105: R = stack([eye x up, up, -eye], dim=-1)."""
    return torch.stack([torch.linalg.cross(eye_vector, up_vector, dim=-1),
                        up_vector, -eye_vector], dim=-1)


def project_camera_rays(image_height, image_width, focal_length_y, focal_length_x):
    out = np.empty((image_height, image_width, 3), dtype=np.float32)
    lib().ref_project_camera_rays(image_height, image_width, float(focal_length_y),
                                  float(focal_length_x), _ptr(out))
    return torch.from_numpy(out)


def transform_rays(rays, eye_vector, up_vector):
    """rays [..., 3] (one camera), eye/up [3] -> world rays, same shape."""
    R = _f32(rotation_from(torch.as_tensor(eye_vector, dtype=torch.float32),
                           torch.as_tensor(up_vector, dtype=torch.float32)).numpy())
    assert R.shape == (3, 3), "oracle transform_rays handles one pose per call"
    r = _f32(torch.as_tensor(rays).numpy())
    out = np.empty_like(r)
    lib().ref_transform_rays(r.size // 3, _ptr(r), _ptr(R), _ptr(out))
    return torch.from_numpy(out)


def bin_rays_dense(bins0, bins1, bins2, origin, rays, depth,
                   min_ray_depth=0.0, max_ray_depth=10.0):
    """Per-pixel (uncompacted) outputs of bin_rays; returns a dict of numpy
    arrays shaped like depth without its last axis, plus `points`."""
    b0, b1, b2 = _f32(torch.as_tensor(bins0).numpy()), _f32(torch.as_tensor(bins1).numpy()), \
        _f32(torch.as_tensor(bins2).numpy())
    r = _f32(torch.as_tensor(rays).numpy())
    d = _f32(torch.as_tensor(depth).numpy())
    assert d.shape[-1] == 1 and r.shape[-1] == 3
    o = _f32(torch.as_tensor(origin).numpy())
    lead = d.shape[:-1]
    if o.ndim == 1:
        n_frames, P = 1, int(np.prod(lead))
        o = o.reshape(1, 3)
    else:
        n_frames = o.shape[0]
        assert lead[0] == n_frames
        P = int(np.prod(lead[1:]))
    r = np.ascontiguousarray(np.broadcast_to(r, lead + (3,)))
    N = n_frames * P
    out = dict(ind0=np.empty(N, np.int64), ind1=np.empty(N, np.int64), ind2=np.empty(N, np.int64),
               ratio0=np.empty(N, np.float32), ratio1=np.empty(N, np.float32),
               ratio2=np.empty(N, np.float32), points=np.empty((N, 3), np.float32),
               valid=np.empty(N, np.uint8))
    lib().ref_bin_rays(_ptr(b0), b0.size, _ptr(b1), b1.size, _ptr(b2), b2.size, n_frames, P,
                       _ptr(o), _ptr(r), _ptr(d), float(min_ray_depth), float(max_ray_depth),
                       _ptr(out["ind0"]), _ptr(out["ind1"]), _ptr(out["ind2"]),
                       _ptr(out["ratio0"]), _ptr(out["ratio1"]), _ptr(out["ratio2"]),
                       _ptr(out["points"]), _ptr(out["valid"]))
    for k in list(out):
        out[k] = out[k].reshape(lead + ((3,) if k == "points" else ()))
    return out


def bin_rays(bins0, bins1, bins2, origin, rays, depth, *features,
             min_ray_depth=0.0, max_ray_depth=10.0):
    """Same return tuple as the reference (compacted with nonzero order)."""
    o = bin_rays_dense(bins0, bins1, bins2, origin, rays, depth,
                       min_ray_depth=min_ray_depth, max_ray_depth=max_ray_depth)
    m = o["valid"].astype(bool)
    res = [torch.from_numpy(o[k][m]) for k in ("ind0", "ind1", "ind2", "ratio0", "ratio1", "ratio2")]
    for f in features:
        f = torch.as_tensor(f)
        res.append(f[torch.from_numpy(m)])
    return tuple(res)


def update_feature_map(ind0, ind1, ind2, ratio0, ratio1, ratio2,
                       features, feature_map, interpolation_weight=1.0, return_touched=False):
    """In place on feature_map (a contiguous fp32 torch tensor or numpy array
    [size0, size1, size2, C]); returns the number of touched voxels (and their
    flat ids when return_touched)."""
    fm = feature_map.numpy() if isinstance(feature_map, torch.Tensor) else feature_map
    assert fm.dtype == np.float32 and fm.flags["C_CONTIGUOUS"] and fm.ndim == 4
    s0, s1, s2, C = fm.shape
    i0 = np.ascontiguousarray(torch.as_tensor(ind0).numpy().astype(np.int64).reshape(-1))
    i1 = np.ascontiguousarray(torch.as_tensor(ind1).numpy().astype(np.int64).reshape(-1))
    i2 = np.ascontiguousarray(torch.as_tensor(ind2).numpy().astype(np.int64).reshape(-1))
    r0, r1, r2 = (_f32(torch.as_tensor(x).numpy()).reshape(-1) for x in (ratio0, ratio1, ratio2))
    f = _f32(torch.as_tensor(features).numpy()).reshape(-1, C)
    n = i0.size
    assert f.shape[0] == n
    touched = np.empty(8 * n if return_touched else 0, np.int64)
    T = lib().ref_update_feature_map(n, _ptr(i0), _ptr(i1), _ptr(i2), _ptr(r0), _ptr(r1), _ptr(r2),
                                     _ptr(f), _ptr(fm), s0, s1, s2, C, float(interpolation_weight),
                                     _ptr(touched) if return_touched else None, touched.size)
    if T < 0:
        raise MemoryError("oracle scratch allocation failed")
    return (T, touched[:T]) if return_touched else T


# --------------------------------------------------------------------------
# layer (mass/nn/base_projection_layer.py), CPU only
# --------------------------------------------------------------------------

def make_bins(origin, n, res):
    """base_projection_layer.py:164-181: n+1 edges from torch.arange."""
    lo = origin - (n + 1) * res / 2
    hi = origin + (n + 1) * res / 2 - 1e-6
    return torch.arange(lo, hi, res, dtype=torch.float32)


class RefProjectionLayer:
    """CPU restatement of BaseProjectionLayer.__init__/reset/update."""

    def __init__(self, camera_height=224, camera_width=224, vertical_fov=90.0,
                 map_height=256, map_width=256, map_depth=64, feature_size=1,
                 origin_y=0.0, origin_x=0.0, origin_z=0.0, grid_resolution=0.05,
                 interpolation_weight=0.5):
        self.camera_height, self.camera_width = camera_height, camera_width
        self.map_height, self.map_width, self.map_depth = map_height, map_width, map_depth
        self.feature_size = feature_size
        self.grid_resolution = grid_resolution
        self.interpolation_weight = interpolation_weight
        focal = camera_height / 2.0 / np.tan(np.radians(vertical_fov) / 2.0)   # :151-152
        self.rays = project_camera_rays(camera_height, camera_width, focal, focal)
        self.data = torch.zeros(map_height, map_width, map_depth, feature_size)
        self.reset(origin_y, origin_x, origin_z)

    def reset(self, origin_y=0.0, origin_x=0.0, origin_z=0.0):
        self.origin_x, self.origin_y, self.origin_z = origin_x, origin_y, origin_z
        self.data.zero_()
        self.bins_x = make_bins(origin_x, self.map_width, self.grid_resolution)
        self.bins_y = make_bins(origin_y, self.map_height, self.grid_resolution)
        self.bins_z = make_bins(origin_z, self.map_depth, self.grid_resolution)

    def update(self, observation, return_touched=False):
        """base_projection_layer.py:282-343 (features already at camera res)."""
        position = torch.as_tensor(observation["position"], dtype=torch.float32)
        yaw = torch.as_tensor(observation["yaw"], dtype=torch.float32)
        elevation = torch.as_tensor(observation["elevation"], dtype=torch.float32)
        depth = torch.as_tensor(observation["depth"], dtype=torch.float32)
        features = torch.as_tensor(observation["features"], dtype=torch.float32)
        features = torch.repeat_interleave(features, self.camera_height // features.shape[0], dim=0)
        features = torch.repeat_interleave(features, self.camera_width // features.shape[1], dim=1)
        rays = transform_rays(self.rays, spherical_to_cartesian(yaw, elevation),
                              spherical_to_cartesian(yaw, elevation + np.pi / 2))
        ix, iy, iz, rx, ry, rz, feats = bin_rays(self.bins_x, self.bins_y, self.bins_z,
                                                 position, rays, depth, features)
        return update_feature_map(iy, ix, iz, ry, rx, rz, feats, self.data,
                                  interpolation_weight=self.interpolation_weight,
                                  return_touched=return_touched)


# --------------------------------------------------------------------------
# matching (mass/utils/experimentation.py:261-287)
# --------------------------------------------------------------------------

def pairwise_l2(f0, f1):
    a, b = _f32(torch.as_tensor(f0).numpy()), _f32(torch.as_tensor(f1).numpy())
    out = np.empty((a.shape[0], b.shape[0]), np.float32)
    lib().ref_pairwise_l2(_ptr(a), a.shape[0], _ptr(b), b.shape[0], a.shape[1], _ptr(out))
    return torch.from_numpy(out)


def match(f0, f1):
    """experimentation.py:284-287: Hungarian on the fp32 cost.  The solver is
    the reference's own third-party dependency (scipy.optimize, unpinned by the
    reference; scipy 1.15.3 in this image)."""
    from scipy.optimize import linear_sum_assignment
    cost = pairwise_l2(f0, f1).numpy()
    rows, cols = linear_sum_assignment(cost)
    return cost, rows, cols


# --------------------------------------------------------------------------
# SemanticProjectionLayer.find (mass/nn/applications/semantic_projection_layer.py:257-362)
# --------------------------------------------------------------------------

def border_boxes(image):
    """Bounding boxes (x, y, w, h) of the borders cv2.findContours(RETR_LIST) would report:
    one per 8-connected component, one per enclosed 4-connected background region (its box
    grown by one pixel).  cv2 is absent here, so the ORDER is canonical (sorted), not cv2's:
    parity unpinned for the order of detections."""
    from scipy import ndimage
    img = np.asarray(image) != 0
    out = []
    lab, _ = ndimage.label(img, structure=np.ones((3, 3)))
    for sl in ndimage.find_objects(lab):
        out.append((sl[1].start, sl[0].start, sl[1].stop - sl[1].start, sl[0].stop - sl[0].start))
    bg, _ = ndimage.label(~np.pad(img, 1), structure=[[0, 1, 0], [1, 1, 1], [0, 1, 0]])
    for k, sl in enumerate(ndimage.find_objects(bg), start=1):
        if k == bg[0, 0]:
            continue
        out.append((sl[1].start - 2, sl[0].start - 2, sl[1].stop - sl[1].start + 2, sl[0].stop - sl[0].start + 2))
    return sorted(out)


def find(data, bins_x, bins_y, bins_z, semantic_category, confidence_threshold=0.2, contour_padding=3,
         contour_threshold=0.0, feature_data=None):
    """The reference's find() on CPU tensors, op for op (:298-357), with border_boxes()
    standing in for cv2.  Returns a list of dicts sorted by box."""
    import torch.nn.functional as functional
    data = torch.as_tensor(data)
    H, W, D, _ = data.shape
    c = semantic_category
    cx = (bins_x[:-1] + bins_x[1:]) / 2
    cy = ((bins_y[:-1] + bins_y[1:]) / 2).flip(-1)
    cz = (bins_z[:-1] + bins_z[1:]) / 2
    yy, xx, zz = torch.meshgrid(torch.arange(H), torch.arange(W), torch.arange(D), indexing='ij')
    coords = torch.stack([cx[xx], cy[yy], cz[zz]], dim=-1)            # map_to_world of integer coordinates
    mask = data[..., c:c + 1]
    smooth = mask.permute(3, 0, 1, 2).unsqueeze(0)
    smooth = functional.avg_pool3d(smooth, contour_padding * 2 + 1, stride=1, padding=contour_padding)
    smooth = smooth.squeeze(0).permute(1, 2, 3, 0)
    image = (smooth > contour_threshold).any(dim=2).numpy().astype(np.uint8)[:, :, 0]
    out = []
    for x, y, w, h in border_boxes(image):
        mask_roi = mask[y:y + h, x:x + w]
        weights = mask_roi / (mask_roi.sum() + 1e-9)
        conf = (mask_roi * weights).sum()
        if conf > confidence_threshold:
            d = dict(box=(x, y, w, h), confidence=float(conf), size=float(mask_roi.sum()),
                     coordinate=(coords[y:y + h, x:x + w] * weights).sum(dim=(0, 1, 2)).numpy())
            if feature_data is not None:
                d["feature"] = (torch.as_tensor(feature_data)[y:y + h, x:x + w] * weights).sum(dim=(0, 1, 2)).numpy()
            out.append(d)
    return out

"""ctypes binding of libmassfuse.so (the C ABI declared in include/massfuse.h).

There is no Python or CPU fallback anywhere in this package: if the shared
library is missing, importing this module raises, and every operator refuses
tensors that are not on a HIP device.
"""
import ctypes
import os

# torch ships its own libamdhip64.so; it has to be in the process BEFORE
# libmassfuse.so is loaded so that both bind to the same HIP runtime (two
# runtimes in one process do not see each other's device pointers).
import torch  # noqa: F401  (import order matters)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MASSFUSE_LIB") or os.path.join(_HERE, "libmassfuse.so")   # override: kernel experiments only

MF_OK, MF_ERR_INVALID, MF_ERR_WORKSPACE, MF_ERR_HIP = 0, -1, -2, -3
FEAT_ONES, FEAT_LABEL_U8, FEAT_LABEL_I32, FEAT_LABEL_I64, FEAT_DENSE_F32 = 0, 1, 2, 3, 4
MODE_SEQUENTIAL, MODE_MERGED = 0, 1
METRIC_L2, METRIC_L2_GEMM, METRIC_COSINE = 0, 1, 2
MAX_FRAMES_PER_CALL = 256
ABI_VERSION = 5
MODE_TILES, MODE_DENSE, MODE_CELLS, MODE_CELLS_AGG = 0, 2, 3, 4
MAP_STATS_PARTS = 2048        # MF_MAP_STATS_PARTS
MAX_MAPS_PER_CALL = 4          # mf_fuse_frame_maps          # mf_fuse_last_mode

c_void_p, c_int32, c_int64, c_float, c_size_t = (ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64,
                                                 ctypes.c_float, ctypes.c_size_t)


class _SizedStruct(ctypes.Structure):
    """Mirror of a C struct whose first member is `uint32_t struct_size` (= sizeof, the ABI handshake
    of include/massfuse.h): filled in on construction."""

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        if not args and "struct_size" not in kw:
            self.struct_size = ctypes.sizeof(self)


class MfGrid(_SizedStruct):
    _fields_ = [("struct_size", ctypes.c_uint32), ("size0", c_int32), ("size1", c_int32), ("size2", c_int32), ("channels", c_int32),
                ("bins_x", c_void_p), ("bins_y", c_void_p), ("bins_z", c_void_p),
                ("n_edges_x", c_int32), ("n_edges_y", c_int32), ("n_edges_z", c_int32),
                ("map", c_void_p)]


class MfFrames(_SizedStruct):
    _fields_ = [("struct_size", ctypes.c_uint32), ("n_frames", c_int32), ("height", c_int32), ("width", c_int32),
                ("cam_rays", c_void_p), ("poses", c_void_p), ("depth", c_void_p), ("feat", c_void_p),
                ("feat_kind", c_int32), ("feat_height", c_int32), ("feat_width", c_int32),
                ("min_depth", c_float), ("max_depth", c_float), ("label_status", c_void_p),
                ("poses_on_host", c_int32)]


# every symbol include/massfuse.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "mf_version": (ctypes.c_int, []),
    "mf_last_error": (ctypes.c_char_p, []),
    "mf_struct_sizes": (ctypes.c_int, [ctypes.POINTER(c_size_t), ctypes.POINTER(c_size_t)]),
    "mf_transform_rays": (ctypes.c_int, [c_void_p, c_int64, c_void_p, c_int32, c_void_p, c_void_p]),
    "mf_bin_rays": (ctypes.c_int, [c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_int32,
                                   c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_int64,
                                   c_float, c_float, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "mf_unproject_bin": (ctypes.c_int, [ctypes.POINTER(MfGrid), ctypes.POINTER(MfFrames),
                                        c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p]),
    "mf_fuse_workspace_bytes": (c_size_t, [ctypes.POINTER(MfGrid), c_int64, c_int32]),
    "mf_fuse_frames": (ctypes.c_int, [ctypes.POINTER(MfGrid), ctypes.POINTER(MfFrames), c_float, c_int32,
                                      c_void_p, c_size_t, c_void_p]),
    "mf_fuse_frames_stage": (ctypes.c_int, [ctypes.POINTER(MfGrid), ctypes.POINTER(MfFrames), c_float, c_int32,
                                            c_void_p, c_size_t, c_void_p]),
    "mf_fuse_frames_commit": (ctypes.c_int, [ctypes.POINTER(MfGrid), ctypes.POINTER(MfFrames), c_float, c_int32,
                                             c_void_p, c_size_t, c_void_p]),
    "mf_fuse_frame_maps": (ctypes.c_int, [ctypes.POINTER(MfGrid), ctypes.POINTER(MfFrames), ctypes.POINTER(c_float), c_int32,
                                          c_int32, ctypes.POINTER(c_void_p), ctypes.POINTER(c_size_t), c_void_p]),
    "mf_update_feature_map": (ctypes.c_int, [ctypes.POINTER(MfGrid), c_int64, c_void_p, c_void_p, c_void_p,
                                             c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_float,
                                             c_void_p, c_size_t, c_void_p]),
    "mf_column_occupied": (ctypes.c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_float,
                                          c_void_p, c_void_p]),
    "mf_amax_z": (ctypes.c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "mf_map_stats": (ctypes.c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "mf_contour_boxes": (ctypes.c_int, [c_void_p, c_int32, c_int32, c_int32, c_void_p, c_int32]),
    "mf_roi_moments": (ctypes.c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_void_p, c_void_p]),
    "mf_profile_enable": (ctypes.c_int, [c_int32]),
    "mf_profile_read": (ctypes.c_int, [c_int32, c_void_p]),
    "mf_fuse_last_mode": (ctypes.c_int, [ctypes.POINTER(MfGrid), c_int64, c_int32, c_void_p, c_void_p]),
    "mf_pairwise_distance": (ctypes.c_int, [c_void_p, c_int32, c_void_p, c_int32, c_int32, c_void_p,
                                            c_int32, c_void_p]),
    "mf_linear_sum_assignment": (ctypes.c_int, [c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
}


class MassFuseError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C mass_amd/csrc` (hipcc, --offload-arch=gfx950). mass_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    with open("/proc/self/maps") as maps:
        runtimes = {line.split()[-1] for line in maps if "libamdhip64" in line}
    if len(runtimes) > 1:
        raise ImportError(f"two HIP runtimes are loaded ({sorted(runtimes)}); import torch before anything "
                          "that links libamdhip64")
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype, fn.argtypes = res, args
    if lib.mf_version() != ABI_VERSION:
        raise ImportError(f"libmassfuse.so ABI {lib.mf_version()} != expected {ABI_VERSION}; rebuild it")
    gs, fs = c_size_t(0), c_size_t(0)
    lib.mf_struct_sizes(ctypes.byref(gs), ctypes.byref(fs))
    if (gs.value, fs.value) != (ctypes.sizeof(MfGrid), ctypes.sizeof(MfFrames)):
        raise ImportError(f"struct layout mismatch: libmassfuse.so has sizeof(mf_grid) = {gs.value}, sizeof(mf_frames) = "
                          f"{fs.value}; mass_amd/_lib.py mirrors {ctypes.sizeof(MfGrid)} and {ctypes.sizeof(MfFrames)} "
                          "(include/massfuse.h and _lib.py are out of step)")
    return lib


lib = _load()


def check(rc):
    """Turn a negative return code into an exception carrying mf_last_error()."""
    if rc < 0:
        msg = lib.mf_last_error().decode("utf-8", "replace")
        if rc == MF_ERR_INVALID:
            raise ValueError(f"massfuse: {msg}")
        raise MassFuseError(f"massfuse error {rc}: {msg}")
    return rc


def require_device(*tensors):
    """The operators run on the GPU only; refuse anything else loudly."""
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("mass_amd operators need tensors on a HIP device (cuda:N); got a "
                               f"{t.device} tensor and there is no CPU fallback")


def ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else None


def current_stream(device):
    import torch
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)

"""CPU-side checks of the C ABI: the library loads, exports every symbol
include/massfuse.h declares, validates arguments before touching the GPU, and
its host-only entry (linear sum assignment) matches scipy."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from mass_amd import _lib


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "massfuse.h")).read()
    return re.findall(r"MF_API\s+[\w\s\*]+?\b(mf_\w+)\s*\(", text)


def test_header_symbols_exported_and_bound():
    names = declared_symbols()
    assert len(names) >= 10 and len(set(names)) == len(names)
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} declared in massfuse.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature in mass_amd/_lib.py"
    assert set(_lib.SIGNATURES) == set(names)


def test_version_matches_header():
    text = open(os.path.join(ROOT, "include", "massfuse.h")).read()
    assert int(re.search(r"#define MF_ABI_VERSION (\d+)", text).group(1)) == _lib.lib.mf_version()


def header_struct_fields(name):
    """Member names of `typedef struct <name> {...}` in include/massfuse.h, in order."""
    text = open(os.path.join(ROOT, "include", "massfuse.h")).read()
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), text, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        for part in decl.split(","):
            fields.append(re.findall(r"\**(\w+)\s*$", part.strip())[0])
    return fields


def integration_stub_fields(cls):
    """The `_fields_` list of class `cls` in INTEGRATION.md's ctypes stub (names and ctypes type names)."""
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    body = re.search(r"class %s\(ctypes\.Structure\):.*?_fields_ = \[(.*?)\]\n" % cls, text, re.S).group(1)
    return re.findall(r'\("(\w+)",\s*ctypes\.(\w+)\)', body)


def test_struct_layout_handshake():
    """ABI 4: mf_struct_sizes equals the ctypes mirrors (checked at import too), the header's members equal the
    mirrors' field for field, and so does the stub INTEGRATION.md shows a maintainer (it had
    fallen a field behind and nothing noticed)."""
    gs, fs = ctypes.c_size_t(), ctypes.c_size_t()
    assert _lib.lib.mf_struct_sizes(ctypes.byref(gs), ctypes.byref(fs)) == _lib.MF_OK
    assert (gs.value, fs.value) == (ctypes.sizeof(_lib.MfGrid), ctypes.sizeof(_lib.MfFrames))
    for cname, cls, doc in (("mf_grid", _lib.MfGrid, "MfGrid"), ("mf_frames", _lib.MfFrames, "MfFrames")):
        mirror = list(cls._fields_)
        assert header_struct_fields(cname) == [n for n, _ in mirror], cname
        stub = [(n, getattr(ctypes, t)) for n, t in integration_stub_fields(doc)]
        assert stub == mirror, f"INTEGRATION.md stub of {doc} is out of step with mass_amd/_lib.py"
    assert "lib.mf_version() == %d" % _lib.ABI_VERSION in open(os.path.join(ROOT, "INTEGRATION.md")).read()


def test_structs_of_another_size_are_refused():
    g = _lib.MfGrid()
    g.size0 = g.size1 = g.size2 = 8
    g.channels = 3
    g.map = 256
    assert _lib.lib.mf_fuse_workspace_bytes(g, 100, 1) > 0
    g.struct_size -= 8                                   # a binding written against a shorter struct
    assert _lib.lib.mf_fuse_workspace_bytes(g, 100, 1) == 0
    assert b"struct_size" in _lib.lib.mf_last_error()
    g.struct_size += 8
    f = _lib.MfFrames()
    f.struct_size = 72                                   # ABI 2's mf_frames (no label_status)
    f.n_frames, f.height, f.width = 1, 4, 4
    f.cam_rays = f.poses = f.depth = 256
    g.bins_x = g.bins_y = g.bins_z = 256
    g.n_edges_x = g.n_edges_y = g.n_edges_z = 9
    assert _lib.lib.mf_fuse_frames(g, f, 0.5, 0, None, 0, None) == _lib.MF_ERR_INVALID
    assert b"mf_frames.struct_size" in _lib.lib.mf_last_error()


def test_argument_validation_needs_no_gpu():
    g = _lib.MfGrid()
    assert _lib.lib.mf_fuse_workspace_bytes(g, 10, 1) == 0
    assert b"map dims" in _lib.lib.mf_last_error()
    g.size0 = g.size1 = g.size2 = 8
    g.channels = 3
    g.map = 256          # never dereferenced on the host
    assert _lib.lib.mf_fuse_workspace_bytes(g, 100, 1) > 0
    assert _lib.lib.mf_fuse_workspace_bytes(g, 100, 1000) == 0      # too many groups
    rc = _lib.lib.mf_fuse_frames(g, None, 0.5, 0, None, 0, None)
    assert rc == _lib.MF_ERR_INVALID
    with pytest.raises(ValueError):
        _lib.check(rc)
    assert _lib.lib.mf_pairwise_distance(None, 2, None, 2, 0, None, 0, None) == _lib.MF_ERR_INVALID
    assert _lib.lib.mf_pairwise_distance(None, 0, None, 2, 4, None, 0, None) == _lib.MF_OK


def test_operators_refuse_cpu_tensors():
    import torch
    from mass_amd.utils import projection as P
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        P.update_feature_map(torch.zeros(1, dtype=torch.int64), torch.zeros(1, dtype=torch.int64),
                             torch.zeros(1, dtype=torch.int64), torch.zeros(1), torch.zeros(1), torch.zeros(1),
                             torch.zeros(1, 2), torch.zeros(4, 4, 4, 2))


def lsa(cost):
    cost = np.ascontiguousarray(cost, np.float64)
    n0, n1 = cost.shape
    k = min(n0, n1)
    rows, cols = np.empty(k, np.int64), np.empty(k, np.int64)
    n = _lib.check(_lib.lib.mf_linear_sum_assignment(cost.ctypes.data, n0, n1, rows.ctypes.data, cols.ctypes.data))
    return rows[:n], cols[:n]


def test_lsa_golden(matchfx):
    tags = sorted({k[:-5] for k in matchfx.files if k.endswith("_rows")})
    assert len(tags) >= 10
    for t in tags:
        cost = matchfx[t + "_cost"]
        rows, cols = lsa(cost)
        assert np.array_equal(rows, matchfx[t + "_rows"]), t
        assert np.array_equal(cols, matchfx[t + "_cols"]), t


def test_lsa_matches_scipy_random_and_ties():
    from scipy.optimize import linear_sum_assignment
    rng = np.random.default_rng(5)
    for trial in range(300):
        n0, n1 = rng.integers(1, 12, 2)
        if trial % 3 == 0:
            cost = rng.integers(0, 4, (n0, n1)).astype(np.float64)      # many ties
        elif trial % 3 == 1:
            cost = rng.standard_normal((n0, n1))
        else:
            cost = rng.random((n0, n1)).astype(np.float32).astype(np.float64)
            cost[rng.random((n0, n1)) < 0.1] = np.inf
        try:
            r, c = linear_sum_assignment(cost)
        except ValueError:
            with pytest.raises(ValueError):
                lsa(cost)
            continue
        gr, gc = lsa(cost)
        assert np.array_equal(gr, r) and np.array_equal(gc, c), (trial, cost)


def test_lsa_rejects_nan_and_handles_empty():
    with pytest.raises(ValueError):
        lsa(np.array([[0.0, np.nan], [1.0, 2.0]]))
    with pytest.raises(ValueError):
        lsa(np.array([[0.0, -np.inf], [1.0, 2.0]]))
    r, c = lsa(np.zeros((0, 3)))
    assert r.size == 0 and c.size == 0


def test_workspace_regrows_when_the_need_grows_by_one_alignment_unit():
    """a need that grows by exactly 256 bytes must not be served from the old buffer."""
    import torch
    from mass_amd.utils.projection import Workspace
    ws = Workspace()
    cpu = torch.device("cpu")
    for need in (102400, 102656, 102912, 1024, 103168):
        p, cap = ws.get(need, cpu)
        assert cap >= need and p.value % 256 == 0
        assert p.value + cap <= ws.buf.data_ptr() + ws.buf.numel()

"""Make the reference's import paths resolve to this package.

    import mass_amd.dropin; mass_amd.dropin.install()
    from mass.nn.applications.semantic_projection_layer import SemanticProjectionLayer   # HIP-backed

Only the modules of the hot path are aliased; anything else under the
reference's ``mass`` package keeps importing from wherever it is installed.  The
reference's application layers import ``slam_rcnn.*`` (a stale package name,
semantic_projection_layer.py:5), so that prefix is aliased too.
"""
import importlib
import sys
import types

_MODULES = {
    "utils.projection": "mass_amd.utils.projection",
    "nn.projection_layer": "mass_amd.nn.projection_layer",
    "nn.base_projection_layer": "mass_amd.nn.base_projection_layer",
    "nn.applications.occupancy_projection_layer": "mass_amd.nn.applications.occupancy_projection_layer",
    "nn.applications.semantic_projection_layer": "mass_amd.nn.applications.semantic_projection_layer",
    "nn.applications.resnet_projection_layer": "mass_amd.nn.applications.resnet_projection_layer",
}


def _ensure_package(name):
    """Return sys.modules[name], creating an empty namespace package if needed."""
    if name not in sys.modules:
        try:
            importlib.import_module(name)
        except Exception:
            pkg = types.ModuleType(name)
            pkg.__path__ = []
            sys.modules[name] = pkg
            if "." in name:
                parent, _, leaf = name.rpartition(".")
                setattr(_ensure_package(parent), leaf, pkg)
    return sys.modules[name]


def install(prefixes=("mass", "slam_rcnn")):
    """Alias <prefix>.<hot-path module> -> mass_amd.<module> for every prefix."""
    for prefix in prefixes:
        for rel, target in _MODULES.items():
            mod = importlib.import_module(target)
            full = f"{prefix}.{rel}"
            parent, _, leaf = full.rpartition(".")
            setattr(_ensure_package(parent), leaf, mod)
            sys.modules[full] = mod
    return sorted(f"{p}.{r}" for p in prefixes for r in _MODULES)

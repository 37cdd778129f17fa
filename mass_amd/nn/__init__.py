"""Projection layers (mirrors mass.nn) and the per-step update of several maps."""
from mass_amd.nn.feature_maps import update_feature_maps  # noqa: F401

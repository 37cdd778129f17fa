"""mass_amd — MI355X-native voxel-map fusion and instance matching for MaSS.

Drop-in for the hot path of brandontrabucco/mass (mass/utils/projection.py,
mass/nn/*projection_layer.py, the matching block of mass/utils/experimentation.py):
the Python call surface is kept, the work runs in hand-written HIP kernels
(libmassfuse.so, C ABI in include/massfuse.h).  See DESIGN.md / INTEGRATION.md.
"""
__version__ = "0.1.0"

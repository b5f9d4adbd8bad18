"""fcaf3d_amd — MI355X-native sparse-voxel hot path of FCAF3D (HIP kernels behind a C ABI)."""
__version__ = '0.1.0'

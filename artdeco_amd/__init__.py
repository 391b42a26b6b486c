"""artdeco_amd -- MI355X-native kernels for ARTDECO's on-the-fly Gaussian-splatting mapper hot path.

The package holds the HIP sources (csrc/), their in-tree build (build.py), the
ctypes binding of the C ABI (_lib.py, include/artdeco_hip.h) and `dropin/`: the
Python modules that carry the reference's own import names (`gsplat`,
`fused_ssim`, `simple_knn`, `diff_gaussian_rasterization`,
`mast3r_slam_backends`, `curope`) so ARTDECO's host code runs unchanged.

    import artdeco_amd; artdeco_amd.install_dropins()   # or: PYTHONPATH=<repo>/artdeco_amd/dropin
"""
import os
import sys

__version__ = "0.1.0"

DROPIN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dropin")


def install_dropins() -> str:
    """Put the drop-in modules first on sys.path (idempotent); returns the directory."""
    if DROPIN_DIR not in sys.path:
        sys.path.insert(0, DROPIN_DIR)
    return DROPIN_DIR

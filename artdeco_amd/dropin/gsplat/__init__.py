"""Drop-in for the `gsplat` package as far as ARTDECO uses it: `gsplat.rendering.rasterization`
(import at Reconstruct/scene/scene_models/h3dgsv3.py:22, call at :664-680), backed by the HIP
kernels of libartdeco_hip.so.  Anything ARTDECO does not call raises NotImplementedError loudly."""
from . import rendering  # noqa: F401
from .rendering import rasterization  # noqa: F401

__version__ = "1.5.0+artdeco_amd"

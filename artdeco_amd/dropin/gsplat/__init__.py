"""Drop-in for the `gsplat` package as far as ARTDECO uses it: `gsplat.rendering.rasterization`
(import at Reconstruct/scene/scene_models/h3dgsv3.py:22, call at :664-680), backed by the HIP
kernels of libartdeco_hip.so.  Anything ARTDECO does not call raises NotImplementedError loudly."""
from . import rendering  # noqa: F401
from artdeco_amd import autoinstall as _autoinstall

_autoinstall.on_dropin_import()  # post-import hook: fused mapper paths on every SceneModel (ARTDECO_AMD_AUTOFUSE=0 disables)
from .rendering import rasterization  # noqa: F401

__version__ = "1.5.0+artdeco_amd"

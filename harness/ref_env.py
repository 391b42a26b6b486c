"""Import ARTDECO's own modules from /root/reference in THIS container (CPU only; the tree is not on the GPU box).

Test / tooling infrastructure (golden generators, tools/make_reference_pins.py): third-party packages that are absent here and
irrelevant to the mapper path (cv2, torchvision, plyfile, lpips, kornia, pypose, open3d, e3nn, cupy ...) are stubbed, the
natives resolve to this package's drop-ins, and `.cuda()` is the identity so the reference's `device="cuda"` habits run on CPU.
"""
from __future__ import annotations

import importlib
import os
import sys
import types

REF = "/root/reference"
SCENE_MODULE = "Reconstruct.scene.scene_models.h3dgsv3"
_STUBS = ("cv2", "torchvision", "torchvision.utils", "plyfile", "lpips", "kornia", "pypose", "open3d", "trimesh", "imageio", "roma",
          "e3nn", "e3nn.o3", "cupy")


class Stub(types.ModuleType):
    """An importable nothing: attribute access yields sub-stubs, calls return the stub."""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        sub = Stub(self.__name__ + "." + name)
        setattr(self, name, sub)
        return sub

    def __call__(self, *a, **k):
        return self


def available() -> bool:
    return os.path.isdir(REF)


def import_scene_module(name: str = SCENE_MODULE):
    """The reference's scene-model module, imported against the drop-ins.  Leaves the stubs in sys.modules (tool processes)."""
    import torch
    import artdeco_amd
    artdeco_amd.install_dropins()
    if REF not in sys.path:
        sys.path.insert(0, REF)
    for m in _STUBS:
        if m not in sys.modules:
            sys.modules[m] = Stub(m)
    if not torch.cuda.is_available():
        torch.Tensor.cuda = lambda self, *a, **k: self
        torch.nn.Module.cuda = lambda self, *a, **k: self
    return importlib.import_module(name)

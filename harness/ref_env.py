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


class TorchNoCuda:
    """`torch` as the reference's methods see it in this CPU-only container: a `device="cuda"` literal means the default device.
    Bound in place of the module-level name `torch` of a reference module (monkeypatch / setattr)."""

    def __getattr__(self, name):
        import torch
        attr = getattr(torch, name)
        if name in ("tensor", "zeros", "ones", "full", "empty", "eye", "arange", "rand", "randn", "zeros_like", "ones_like"):
            def factory(*a, **k):
                if isinstance(k.get("device"), str) and k["device"].startswith("cuda"):
                    k.pop("device")
                return attr(*a, **k)
            return factory
        return attr


def cpu_natives():
    """(rasterization, fused_ssim, adamUpdate, adamUpdateBasic, scatter_max) bound to the CPU oracles, with the reference's
    call signatures -- what the real SceneModel / harness mirror run on when they are executed in this container."""
    import numpy as np
    import torch
    from oracle import adam_oracle, gsplat_oracle, scatter_oracle, ssim_oracle

    def rasterization(means, quats, scales, opacities, colors, viewmats, Ks, width, height, render_mode, rasterize_mode, absgrad, packed,
                      sh_degree, eps2d):
        r, a, meta = gsplat_oracle.rasterization(means, quats, scales, opacities, colors, viewmats[0], Ks[0], width, height,
                                                 sh_degree=sh_degree, eps2d=eps2d, grad_dtype=torch.float32)
        return r[None], a[None], {"radii": meta["radii"][None]}

    def fused_ssim(img1, img2, padding="same", train=True):
        return ssim_oracle.fused_ssim_oracle(img1, img2, padding)

    def adam_update(param, grad, exp_avg, exp_avg_sq, visible, lr, b1, b2, eps, N, M):
        lr_np = lr.detach().numpy() if torch.is_tensor(lr) else np.float32(lr)
        p, m, v = adam_oracle.adam_update_oracle(param.detach().numpy(), grad.numpy(), exp_avg.numpy(), exp_avg_sq.numpy(),
                                                 visible.numpy(), lr_np, b1, b2, eps, N, M)
        param.data.copy_(torch.from_numpy(p)); exp_avg.copy_(torch.from_numpy(m)); exp_avg_sq.copy_(torch.from_numpy(v))

    def adam_update_basic(param, grad, exp_avg, exp_avg_sq, lr, b1, b2, eps):
        p, m, v = adam_oracle.adam_update_basic_oracle(param.detach().numpy(), grad.numpy(), exp_avg.numpy(), exp_avg_sq.numpy(), lr, b1, b2, eps)
        param.data.copy_(torch.from_numpy(p)); exp_avg.copy_(torch.from_numpy(m)); exp_avg_sq.copy_(torch.from_numpy(v))

    def scatter_max(src, index, *a, **k):
        out, arg = scatter_oracle.scatter_arg(src.numpy(), index.numpy())
        return torch.from_numpy(out), torch.from_numpy(arg)

    return rasterization, fused_ssim, adam_update, adam_update_basic, scatter_max

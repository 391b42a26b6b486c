"""Drop-in for the `fused_ssim` package: `fused_ssim(img1, img2, padding="same", train=True) -> 0-dim tensor`,
`allowed_padding`, and the map-level autograd node `FusedSSIMMap` (binding surface of
Reconstruct/submodules/fused-ssim/fused_ssim/__init__.py; call sites Reconstruct/scene/scene_models/h3dgsv3.py:441
training loss, :545 evaluation).  Gradient with respect to img1 only, like the reference.

Not a transcription of the reference wrapper: the scalar entry point is its own autograd node that goes straight from the
two images to mean(SSIM) and back.  Its backward uses the C ABI's uniform-gradient form (dL_dmap = NULL, dL_scalar) -- the
gradient of a mean is one number, so no [B,CH,H,W] gradient map is materialised or read (-4 B/px.ch against the
map-then-mean route) -- and scales the result by the incoming 0-dim gradient on the device, so there is no host read.
"""
from __future__ import annotations

import torch

from artdeco_amd import _lib
from artdeco_amd import autoinstall as _autoinstall

_autoinstall.on_dropin_import()  # post-import hook: fused mapper paths on every SceneModel (ARTDECO_AMD_AUTOFUSE=0 disables)

allowed_padding = ["same", "valid"]
_C1, _C2 = 0.01 ** 2, 0.03 ** 2   # (k1 L)^2, (k2 L)^2 with L = 1, the constants the reference passes
_CROP = 5                         # "valid" = the 11-tap window fully inside the image


def _images(img1: torch.Tensor, img2: torch.Tensor):
    _lib.require_cuda(img1, img2)
    if img1.dim() != 4 or img1.shape != img2.shape:
        raise ValueError(f"fused_ssim expects two [B,CH,H,W] tensors of equal shape, got {tuple(img1.shape)} / {tuple(img2.shape)}")
    if img1.dtype != torch.float32 or img2.dtype != torch.float32:
        raise TypeError("fused_ssim operates on float32 tensors")
    return img1.detach().contiguous(), img2.detach().contiguous()


def _forward(a: torch.Tensor, b: torch.Tensor, C1: float, C2: float, train: bool):
    lib = _lib.load()
    B, CH, H, W = a.shape
    with _lib.on_device(a.device):
        ssim_map = torch.empty_like(a)
        dmaps = tuple(torch.empty_like(a) for _ in range(3)) if train else (None, None, None)
        rc = lib.adk_fused_ssim_fwd(a.data_ptr(), b.data_ptr(), B, CH, H, W, float(C1), float(C2), ssim_map.data_ptr(),
                                    _lib.ptr(dmaps[0]), _lib.ptr(dmaps[1]), _lib.ptr(dmaps[2]), _lib.stream_of(a))
    _lib.check(rc, "adk_fused_ssim_fwd")
    return ssim_map, dmaps


def _backward(a, b, dmaps, dL_dmap, dL_scalar: float):
    lib = _lib.load()
    B, CH, H, W = a.shape
    with _lib.on_device(a.device):
        out = torch.empty_like(a)
        rc = lib.adk_fused_ssim_bwd(a.data_ptr(), b.data_ptr(), _lib.ptr(dL_dmap), float(dL_scalar), dmaps[0].data_ptr(),
                                    dmaps[1].data_ptr(), dmaps[2].data_ptr(), B, CH, H, W, out.data_ptr(), _lib.stream_of(a))
    _lib.check(rc, "adk_fused_ssim_bwd")
    return out


class _SSIMMean(torch.autograd.Function):
    """(img1, img2) -> mean SSIM over the ("same": whole, "valid": cropped) map."""

    @staticmethod
    def forward(ctx, img1, img2, padding, train):
        a, b = _images(img1, img2)
        ssim_map, dmaps = _forward(a, b, _C1, _C2, train)
        region = ssim_map if padding == "same" else ssim_map[:, :, _CROP:-_CROP, _CROP:-_CROP]
        ctx.padding, ctx.count, ctx.train = padding, region.numel(), train
        if train:
            ctx.save_for_backward(a, b, *dmaps)
        return region.mean()

    @staticmethod
    def backward(ctx, v_mean):
        if not ctx.train:
            raise RuntimeError("fused_ssim(..., train=False) keeps no derivative maps; call it with train=True to back-propagate")
        a, b, d0, d1, d2 = ctx.saved_tensors
        if ctx.padding == "same":
            grad = _backward(a, b, (d0, d1, d2), None, 1.0 / ctx.count)       # uniform gradient: no map read
        else:
            dL = torch.zeros_like(a)
            dL[:, :, _CROP:-_CROP, _CROP:-_CROP] = 1.0 / ctx.count
            grad = _backward(a, b, (d0, d1, d2), dL, 0.0)
        return grad.mul_(v_mean), None, None, None


class FusedSSIMMap(torch.autograd.Function):
    """The map-level node under the reference's name and call signature, for code that wants the per-pixel SSIM:
    FusedSSIMMap.apply(C1, C2, img1, img2, padding="same", train=True) -> ssim map."""

    @staticmethod
    def forward(ctx, C1, C2, img1, img2, padding="same", train=True):
        a, b = _images(img1, img2)
        ssim_map, dmaps = _forward(a, b, C1, C2, train)
        ctx.padding, ctx.train = padding, train
        if train:
            ctx.save_for_backward(a, b, *dmaps)
        return ssim_map if padding == "same" else ssim_map[:, :, _CROP:-_CROP, _CROP:-_CROP]

    @staticmethod
    def backward(ctx, v_map):
        if not ctx.train:
            raise RuntimeError("FusedSSIMMap(train=False) keeps no derivative maps")
        a, b, d0, d1, d2 = ctx.saved_tensors
        if ctx.padding == "same":
            dL = v_map.contiguous()
        else:
            dL = torch.zeros_like(a)
            dL[:, :, _CROP:-_CROP, _CROP:-_CROP] = v_map
        return None, None, _backward(a, b, (d0, d1, d2), dL, 0.0), None, None, None


def fused_ssim(img1, img2, padding="same", train=True):
    if padding not in allowed_padding:
        raise ValueError(f"padding must be one of {allowed_padding}")
    return _SSIMMean.apply(img1, img2, padding, bool(train))

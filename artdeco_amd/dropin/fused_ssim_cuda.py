"""Drop-in for the `fused_ssim_cuda` extension module (fused-ssim/ext.cpp:4-7).

Exposes `fusedssim` and `fusedssim_backward` with the reference signatures
(ssim.cu:434-440, :481-491) on top of the HIP kernels in artdeco_amd/csrc/ssim.hip.
"""
from __future__ import annotations

import torch

from artdeco_amd import _lib


def _check(img1: torch.Tensor, img2: torch.Tensor) -> None:
    _lib.require_cuda(img1, img2)
    if img1.dim() != 4 or img1.shape != img2.shape:
        raise ValueError(f"fused_ssim expects two [B,CH,H,W] tensors of equal shape, got {tuple(img1.shape)} / {tuple(img2.shape)}")
    if img1.dtype != torch.float32 or img2.dtype != torch.float32:
        raise TypeError("fused_ssim operates on float32 tensors")


def fusedssim(C1: float, C2: float, img1: torch.Tensor, img2: torch.Tensor, train: bool):
    """-> (ssim_map, dm_dmu1, dm_dsigma1_sq, dm_dsigma12); the last three are empty when train=False."""
    _check(img1, img2)
    lib = _lib.load()
    img1c, img2c = img1.contiguous(), img2.contiguous()
    B, CH, H, W = img1c.shape
    with torch.cuda.device(img1c.device):
        ssim_map = torch.empty_like(img1c)
        if train:
            dm_dmu1 = torch.empty_like(img1c)
            dm_dsigma1_sq = torch.empty_like(img1c)
            dm_dsigma12 = torch.empty_like(img1c)
        else:
            dm_dmu1 = torch.empty(0, device=img1c.device, dtype=img1c.dtype)
            dm_dsigma1_sq = torch.empty(0, device=img1c.device, dtype=img1c.dtype)
            dm_dsigma12 = torch.empty(0, device=img1c.device, dtype=img1c.dtype)
        rc = lib.adk_fused_ssim_fwd(
            img1c.data_ptr(), img2c.data_ptr(), B, CH, H, W, float(C1), float(C2), ssim_map.data_ptr(),
            dm_dmu1.data_ptr() if train else None, dm_dsigma1_sq.data_ptr() if train else None,
            dm_dsigma12.data_ptr() if train else None, _lib.stream_of(img1c))
    _lib.check(rc, "adk_fused_ssim_fwd")
    return ssim_map, dm_dmu1, dm_dsigma1_sq, dm_dsigma12


def fusedssim_backward(C1: float, C2: float, img1: torch.Tensor, img2: torch.Tensor, dL_dmap: torch.Tensor,
                       dm_dmu1: torch.Tensor, dm_dsigma1_sq: torch.Tensor, dm_dsigma12: torch.Tensor):
    """-> dL/d(img1), [B,CH,H,W]."""
    _check(img1, img2)
    lib = _lib.load()
    img1c, img2c = img1.contiguous(), img2.contiguous()
    B, CH, H, W = img1c.shape
    if dm_dmu1.numel() != img1c.numel():
        raise ValueError("fusedssim_backward needs the derivative maps of a train=True forward")
    with torch.cuda.device(img1c.device):
        dL = dL_dmap.contiguous()
        out = torch.empty_like(img1c)
        rc = lib.adk_fused_ssim_bwd(
            img1c.data_ptr(), img2c.data_ptr(), dL.data_ptr(), 0.0, dm_dmu1.contiguous().data_ptr(),
            dm_dsigma1_sq.contiguous().data_ptr(), dm_dsigma12.contiguous().data_ptr(), B, CH, H, W,
            out.data_ptr(), _lib.stream_of(img1c))
    _lib.check(rc, "adk_fused_ssim_bwd")
    return out

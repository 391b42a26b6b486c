"""Drop-in for the `mast3r_slam_backends` extension (VSLAM/backend, pybind at src/gn.cpp:116-122).

Implemented on HIP: iter_proj (gn.cpp:84-99) and refine_matches (gn.cpp:101-114), the two entry
points on the frontend hot path (VSLAM/utils_matching.py:152-159, :171-179).  The Gauss-Newton
global optimiser entry points (gauss_newton_points / rays / calib) belong to the backend process
and are next-tier: they raise NotImplementedError rather than fall back.
"""
from __future__ import annotations

import torch

from artdeco_amd import _lib


def _contig(t, name):
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")  # CHECK_CONTIGUOUS, gn.h:5
    return t


def iter_proj(rays_img_with_grad, pts_3d_norm, p_init, max_iter, lambda_init, cost_thresh):
    """-> [p_new float [b,n,2], converged bool [b,n]]"""
    _lib.require_cuda(rays_img_with_grad, pts_3d_norm, p_init)
    rays, pts, p0 = (_contig(rays_img_with_grad, "rays_img_with_grad"), _contig(pts_3d_norm, "pts_3d_norm"),
                     _contig(p_init, "p_init"))
    if rays.dtype != torch.float32 or pts.dtype != torch.float32 or p0.dtype != torch.float32:
        raise TypeError("iter_proj expects float32 tensors")
    b, h, w, c = rays.shape
    if c != 9:
        raise ValueError("rays_img_with_grad must be [b,h,w,9]")
    n = p0.shape[1]
    if pts.shape != (b, n, 3) or p0.shape != (b, n, 2):
        raise ValueError("pts_3d_norm [b,n,3] and p_init [b,n,2] expected")
    lib = _lib.load()
    with torch.cuda.device(rays.device):
        p_new = torch.empty(b, n, 2, dtype=torch.float32, device=rays.device)
        conv = torch.empty(b, n, dtype=torch.bool, device=rays.device)
        rc = lib.adk_iter_proj(rays.data_ptr(), pts.data_ptr(), p0.data_ptr(), b, h, w, n, int(max_iter),
                               float(lambda_init), float(cost_thresh), p_new.data_ptr(), conv.data_ptr(),
                               _lib.stream_of(rays))
    _lib.check(rc, "adk_iter_proj")
    return [p_new, conv]


def refine_matches(D11, D21, p1, window_size, dilation_max):
    """-> [p1_new int64 [b,n,2]]  (window_size is the search radius, gn.cpp:105)"""
    _lib.require_cuda(D11, D21, p1)
    D11, D21, p1 = _contig(D11, "D11"), _contig(D21, "D21"), _contig(p1, "p1")
    if D11.dtype != D21.dtype or D11.dtype not in (torch.float16, torch.float32):
        raise TypeError("refine_matches expects float16 or float32 descriptors of one dtype")
    if p1.dtype != torch.int64:
        raise TypeError("p1 must be int64")
    b, h, w, f = D11.shape
    n = p1.shape[1]
    if D21.shape != (b, n, f) or p1.shape != (b, n, 2):
        raise ValueError("D21 [b,n,f] and p1 [b,n,2] expected")
    lib = _lib.load()
    with torch.cuda.device(D11.device):
        out = torch.empty_like(p1)
        rc = lib.adk_refine_matches(D11.data_ptr(), D21.data_ptr(), p1.data_ptr(), 0 if D11.dtype == torch.float16 else 1,
                                    b, h, w, n, f, int(window_size), int(dilation_max), out.data_ptr(),
                                    _lib.stream_of(D11))
    _lib.check(rc, "adk_refine_matches")
    return [out]


def _next_tier(name):
    def fn(*a, **k):
        raise NotImplementedError(f"mast3r_slam_backends.{name}: Gauss-Newton global optimiser is not part of the "
                                  "mapper hot path implemented by artdeco_amd (SURVEY.md 8f-3)")
    fn.__name__ = name
    return fn


gauss_newton_points = _next_tier("gauss_newton_points")
gauss_newton_rays = _next_tier("gauss_newton_rays")
gauss_newton_calib = _next_tier("gauss_newton_calib")

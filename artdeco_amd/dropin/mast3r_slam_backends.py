"""Drop-in for the `mast3r_slam_backends` extension (VSLAM/backend, pybind at src/gn.cpp:116-122).

Implemented on HIP: iter_proj (gn.cpp:84-99) and refine_matches (gn.cpp:101-114), the two entry
points on the frontend hot path (VSLAM/utils_matching.py:152-159, :171-179), and the Sim(3)
Gauss-Newton global optimiser of the backend process (gauss_newton_points / rays / calib,
gn.cpp:3-82) with the normal equations solved on the device.
"""
from __future__ import annotations

import torch

from artdeco_amd import _lib
from artdeco_amd import autoinstall as _autoinstall

_autoinstall.on_dropin_import()  # post-import hook: fused mapper paths on every SceneModel (ARTDECO_AMD_AUTOFUSE=0 disables)


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


_NUM_FIX = 1  # gn_kernels.cu:760, :1133, :1582: the first keyframe of the graph is held fixed


def _gauss_newton(kind, Twc, Xs, Cs, K, ii, jj, idx_ii2jj, valid_match, Q, height, width, pixel_border, z_eps,
                  sigma_a, sigma_b, C_thresh, Q_thresh, max_iter, delta_thresh, debug_blocks=False):
    names = ("Twc", "Xs", "Cs", "ii", "jj", "idx_ii2jj", "valid_match", "Q")
    tens = (Twc, Xs, Cs, ii, jj, idx_ii2jj, valid_match, Q)
    _lib.require_cuda(*tens)
    for t, n in zip(tens, names):
        _contig(t, n)
    if Twc.dtype != torch.float32 or Xs.dtype != torch.float32 or Cs.dtype != torch.float32 or Q.dtype != torch.float32:
        raise TypeError("gauss_newton: Twc, Xs, Cs, Q must be float32")
    if ii.dtype != torch.int64 or jj.dtype != torch.int64 or idx_ii2jj.dtype != torch.int64 or valid_match.dtype != torch.bool:
        raise TypeError("gauss_newton: ii, jj, idx_ii2jj int64 and valid_match bool expected")
    P, n = Xs.shape[0], Xs.shape[1]
    E = ii.shape[0]
    if Twc.shape != (P, 8) or Xs.shape != (P, n, 3) or Cs.numel() != P * n:
        raise ValueError("gauss_newton: Twc [P,8], Xs [P,n,3], Cs [P,n,1] expected")
    if jj.shape != (E,) or idx_ii2jj.shape != (E, n) or valid_match.numel() != E * n or Q.numel() != E * n:
        raise ValueError("gauss_newton: ii/jj [E], idx_ii2jj [E,n], valid_match [E,n,1], Q [E,n,1] expected")
    dev = Twc.device
    lib = _lib.load()
    with torch.cuda.device(dev):
        # keyframe id -> position in the pose arrays (get_unique_kf_idx + create_inds, gn_kernels.cu:157-169)
        unique_kf_idx = torch.unique(torch.cat([ii, jj]), sorted=True)
        ii_edge = torch.searchsorted(unique_kf_idx, ii).contiguous()
        jj_edge = torch.searchsorted(unique_kf_idx, jj).contiguous()
        dx = torch.zeros(max(P - _NUM_FIX, 0), 7, dtype=torch.float32, device=dev)
        Hs = gs = None
        if debug_blocks:
            Hs = torch.zeros(4, E, 7, 7, dtype=torch.float32, device=dev)
            gs = torch.zeros(2, E, 7, dtype=torch.float32, device=dev)
        Kc = None
        if kind == 2:
            _lib.require_cuda(K)
            Kc = K.detach().contiguous().float()
        ws = torch.empty(int(lib.adk_gn_workspace_bytes(P, E, n)), dtype=torch.uint8, device=dev)
        rc = lib.adk_gauss_newton(kind, P, E, n, Twc.data_ptr(), Xs.data_ptr(), Cs.data_ptr(), _lib.ptr(Kc), ii_edge.data_ptr(),
                                  jj_edge.data_ptr(), idx_ii2jj.data_ptr(), valid_match.data_ptr(), Q.data_ptr(), int(height),
                                  int(width), int(pixel_border), float(z_eps), float(sigma_a), float(sigma_b), float(C_thresh),
                                  float(Q_thresh), int(max_iter), float(delta_thresh), _NUM_FIX, dx.data_ptr(), _lib.ptr(Hs),
                                  _lib.ptr(gs), ws.data_ptr(), ws.numel(), _lib.stream_of(Twc))
    _lib.check(rc, "adk_gauss_newton")
    return [dx, Hs, gs] if debug_blocks else [dx]


def gauss_newton_points(Twc, Xs, Cs, ii, jj, idx_ii2jj, valid_match, Q, sigma_point, C_thresh, Q_thresh, max_iter,
                        delta_thresh):
    """gn.cpp:3-27.  Twc [P,8] is updated in place; returns [dx] (the last step, as the reference does)."""
    return _gauss_newton(0, Twc, Xs, Cs, None, ii, jj, idx_ii2jj, valid_match, Q, 0, 0, 0, 0.0, sigma_point, 1.0, C_thresh,
                         Q_thresh, max_iter, delta_thresh)


def gauss_newton_rays(Twc, Xs, Cs, ii, jj, idx_ii2jj, valid_match, Q, sigma_ray, sigma_dist, C_thresh, Q_thresh, max_iter,
                      delta_thresh):
    """gn.cpp:29-54; caller VSLAM/mast3r_slam/global_opt.py:158-173."""
    return _gauss_newton(1, Twc, Xs, Cs, None, ii, jj, idx_ii2jj, valid_match, Q, 0, 0, 0, 0.0, sigma_ray, sigma_dist, C_thresh,
                         Q_thresh, max_iter, delta_thresh)


def gauss_newton_calib(Twc, Xs, Cs, K, ii, jj, idx_ii2jj, valid_match, Q, height, width, pixel_border, z_eps, sigma_pixel,
                       sigma_depth, C_thresh, Q_thresh, max_iter, delta_thresh):
    """gn.cpp:56-82; caller VSLAM/mast3r_slam/global_opt.py:208-228."""
    return _gauss_newton(2, Twc, Xs, Cs, K, ii, jj, idx_ii2jj, valid_match, Q, height, width, pixel_border, z_eps, sigma_pixel,
                         sigma_depth, C_thresh, Q_thresh, max_iter, delta_thresh)

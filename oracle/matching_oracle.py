"""CPU oracle for mast3r_slam_backends.iter_proj / refine_matches -- TEST INFRASTRUCTURE only.

Restates, operation for operation, VSLAM/backend/src/matching_kernels.cu:
  iter_proj_kernel       :119-275   (per-pixel Levenberg-Marquardt on the bilinear ray image)
  refine_matches_kernel  :25-81     (coarse-to-fine windowed descriptor argmax, scalar_t accumulation)
vectorised over pixels with numpy.  Precision follows the C++ exactly: float32 everywhere except
where the source writes a double literal (`1.0-du`, `1.0/r_norm`, `1.0/(...)`, `lambda *= 0.1`), which
promotes that sub-expression to float64 before the store to a float.  No FMA contraction (nvcc
would contract; the HIP kernel is built with -ffp-contract=off, so kernel == oracle bit for bit).

Parity: UNPINNED by the reference (CUDA-only source, no test or golden vector in the tree).
"""
from __future__ import annotations

import numpy as np

f32, f64 = np.float32, np.float64


def _bilinear_weights(u, v):
    u11 = np.floor(u).astype(np.int32)
    v11 = np.floor(v).astype(np.int32)
    du = u - u11.astype(f32)
    dv = v - v11.astype(f32)
    w11 = du * dv
    w12 = ((1.0 - du.astype(f64)) * dv.astype(f64)).astype(f32)
    w21 = (du.astype(f64) * (1.0 - dv.astype(f64))).astype(f32)
    w22 = ((1.0 - du.astype(f64)) * (1.0 - dv.astype(f64))).astype(f32)
    return u11, v11, w11, w12, w21, w22


def _interp(img, u11, v11, w11, w12, w21, w22, lo, hi):
    # NOTE the reference pairs w11 with the (v+1,u+1) pixel etc. (:161-170) -- kept literally
    r11 = img[v11 + 1, u11 + 1, lo:hi]
    r12 = img[v11 + 1, u11, lo:hi]
    r21 = img[v11, u11 + 1, lo:hi]
    r22 = img[v11, u11, lo:hi]
    return ((w11[:, None] * r11 + w12[:, None] * r12) + w21[:, None] * r21) + w22[:, None] * r22


def _normalize_err(r, tgt):
    r_norm = np.sqrt((r[:, 0] * r[:, 0] + r[:, 1] * r[:, 1]) + r[:, 2] * r[:, 2])
    r_norm_inv = (1.0 / r_norm.astype(f64)).astype(f32)
    r = r * r_norm_inv[:, None]
    err = r - tgt
    cost = (err[:, 0] * err[:, 0] + err[:, 1] * err[:, 1]) + err[:, 2] * err[:, 2]
    return err, cost


def iter_proj_oracle(rays_img_with_grad, pts_3d_norm, p_init, max_iter, lambda_init, cost_thresh):
    """rays [b,h,w,9] f32, pts [b,n,3] f32, p_init [b,n,2] f32 -> (p_new [b,n,2] f32, converged [b,n] bool)."""
    rays = np.ascontiguousarray(rays_img_with_grad, dtype=f32)
    pts = np.ascontiguousarray(pts_3d_norm, dtype=f32)
    p0 = np.ascontiguousarray(p_init, dtype=f32)
    B, h, w, _ = rays.shape
    n = pts.shape[1]
    p_new = np.zeros((B, n, 2), f32)
    conv = np.zeros((B, n), bool)
    for b in range(B):
        img, tgt = rays[b], pts[b]
        u = np.minimum(np.maximum(p0[b, :, 0], f32(1)), f32(w - 2))
        v = np.minimum(np.maximum(p0[b, :, 1], f32(1)), f32(h - 2))
        lam = np.full(n, f32(lambda_init), f32)
        c = np.zeros(n, bool)
        for _ in range(max_iter):
            u11, v11, w11, w12, w21, w22 = _bilinear_weights(u, v)
            r = _interp(img, u11, v11, w11, w12, w21, w22, 0, 3)
            gx = _interp(img, u11, v11, w11, w12, w21, w22, 3, 6)
            gy = _interp(img, u11, v11, w11, w12, w21, w22, 6, 9)
            err, cost = _normalize_err(r, tgt)
            dot = lambda a, bb: (a[:, 0] * bb[:, 0] + a[:, 1] * bb[:, 1]) + a[:, 2] * bb[:, 2]
            A00, A01, A11 = dot(gx, gx), dot(gx, gy), dot(gy, gy)
            b0, b1 = -dot(err, gx), -dot(err, gy)
            A00 = A00 + lam
            A11 = A11 + lam
            with np.errstate(divide="ignore", invalid="ignore"):
                det_inv = (1.0 / (A00 * A11 - A01 * A01).astype(f64)).astype(f32)
                delta_u = det_inv * (A11 * b0 - A01 * b1)
                delta_v = det_inv * ((-A01) * b0 + A00 * b1)
            u_new = np.minimum(np.maximum(u + delta_u, f32(1)), f32(w - 2))
            v_new = np.minimum(np.maximum(v + delta_v, f32(1)), f32(h - 2))
            # fmin/fmax return the non-NaN operand; emulate for the (degenerate) NaN case
            u_new = np.where(np.isnan(u_new), f32(w - 2), u_new).astype(f32)
            v_new = np.where(np.isnan(v_new), f32(h - 2), v_new).astype(f32)
            u11, v11, w11, w12, w21, w22 = _bilinear_weights(u_new, v_new)
            r2 = _interp(img, u11, v11, w11, w12, w21, w22, 0, 3)
            _, new_cost = _normalize_err(r2, tgt)
            better = new_cost < cost
            u = np.where(better, u_new, u)
            v = np.where(better, v_new, v)
            lam = np.where(better, (lam.astype(f64) * 0.1).astype(f32), (lam.astype(f64) * 10.0).astype(f32))
            c = np.where(better, new_cost < f32(cost_thresh), cost < f32(cost_thresh))
        p_new[b, :, 0], p_new[b, :, 1], conv[b] = u, v, c
    return p_new, conv


def refine_matches_oracle(D11, D21, p1, radius, dilation_max):
    """D11 [b,h,w,f], D21 [b,n,f] (float16 or float32), p1 [b,n,2] int64 -> p1_new [b,n,2] int64.
    Score accumulates in the arrays' own dtype: prod rounded to T, then sum rounded to T (:60-63)."""
    T = D11.dtype.type
    assert D21.dtype == D11.dtype and T in (np.float16, np.float32)
    B, h, w, fdim = D11.shape
    n = D21.shape[1]
    out = np.zeros((B, n, 2), np.int64)
    min_pos = np.finfo(T).tiny  # numeric_limits<T>::min(): smallest positive NORMAL (:47)
    for b in range(B):
        u0 = p1[b, :, 0].astype(np.int64).copy()
        v0 = p1[b, :, 1].astype(np.int64).copy()
        max_score = np.full(n, min_pos, T)
        u_new, v_new = u0.copy(), v0.copy()
        q = D21[b]
        for d in range(dilation_max, 0, -1):
            rd = radius * d
            diam = 2 * rd + 1
            for i in range(0, diam, d):
                for j in range(0, diam, d):
                    u = u0 - rd + i
                    v = v0 - rd + j
                    inside = (v >= 0) & (v < h) & (u >= 0) & (u < w)
                    c = D11[b, np.clip(v, 0, h - 1), np.clip(u, 0, w - 1)]
                    score = np.zeros(n, T)
                    for k in range(fdim):
                        score = (score + (q[:, k] * c[:, k]).astype(T)).astype(T)
                    better = inside & (score > max_score)
                    max_score = np.where(better, score, max_score)
                    u_new = np.where(better, u, u_new)
                    v_new = np.where(better, v, v_new)
            u0, v0 = u_new.copy(), v_new.copy()
        out[b, :, 0], out[b, :, 1] = u_new, v_new
    return out

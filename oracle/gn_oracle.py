"""ORACLE (test infrastructure only): Sim(3) Gauss-Newton global optimiser, numpy restatement.

Follows VSLAM/backend/src/gn_kernels.cu literally -- the per-factor 14x14 Hessian of [pose i, pose j] with
J_j = apply_Sim3_adj_inv(T_i; J0), J_i = -J_j (:560-680 points, :920-1090 rays, :1346-1480 calib), the block layout
Hs[4,E,7,7] / gs[2,E,7] (:713-745), the sparse assembly that drops the fixed pose (SparseBlock, :58-160), the
`dx = -solve` / zero-on-failure rule, retrSim3 / expSim3 with their series branches (:302-413), and the
|dx| < delta_thresh stop (:801-806) -- but vectorised over points and in float64 (the reference accumulates in
float32; the comparison tolerance covers that).  The reference extension cannot be built here (CUDA-only source,
Eigen submodule absent) and ships no test or golden vector for these entry points.  PINNED (rays, calib) on a single
factor: tests/golden/gn_factor_{rays,calib}.npz hold every iteration's (tau, cost) and the final pose of the (i, j)
factor of a two-keyframe graph solved by the reference's OWN Python code for the same residuals
(CameraTracker.opt_pose_ray_dist_sim3 / opt_pose_calib_sim3, VSLAM/CameraTracker.py:242-396, run on CPU by
tests/golden/make_golden_gn_factor.py), and tests/test_gn.py requires one Gauss-Newton step of this oracle to be that
step.  The `points` kind and the multi-factor assembly have no Python counterpart in the reference: unpinned; they are
checked against finite differences of their own residuals and against recovery of known ground-truth poses.
"""
from __future__ import annotations

import numpy as np


# ---------------------------------------------------------------- Sim(3) helpers (quaternions xyzw)
def quat_comp(a, b):
    return np.array([a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1],
                     a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0],
                     a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3],
                     a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2]], dtype=a.dtype)


def act_so3(q, X):
    """X [...,3] rotated by q (gn_kernels.cu:196-206)."""
    qv = q[:3]
    uv = 2.0 * np.cross(np.broadcast_to(qv, X.shape), X)
    return X + q[3] * uv + np.cross(np.broadcast_to(qv, X.shape), uv)


def rel_sim3(ti, qi, si, tj, qj, sj):
    si_inv = 1.0 / si
    qi_inv = np.array([-qi[0], -qi[1], -qi[2], qi[3]], dtype=qi.dtype)
    return act_so3(qi_inv, tj - ti) * si_inv, quat_comp(qi_inv, qj), si_inv * sj


def apply_sim3_adj_inv(t, q, s, X):
    """X [...,7] -> Y [...,7] (gn_kernels.cu:273-299)."""
    s_inv = 1.0 / s
    Ra = act_so3(q, X[..., 0:3])
    Y = np.empty_like(X)
    Y[..., 0:3] = s_inv * Ra
    Y[..., 3:6] = act_so3(q, X[..., 3:6]) + s_inv * np.cross(np.broadcast_to(t, Ra.shape), Ra)
    Y[..., 6] = X[..., 6] + s_inv * (Ra @ t)
    return Y


def exp_sim3(xi):
    """float32 in, float32 out; branches exactly as gn_kernels.cu:302-385."""
    f = np.float32
    xi = xi.astype(f)
    tau, phi, sigma = xi[0:3].copy(), xi[3:6], xi[6]
    EPS = f(1e-6)
    scale = np.exp(sigma, dtype=f)
    theta_sq = f(phi @ phi)
    if theta_sq < EPS:
        p4 = theta_sq * theta_sq
        imag = f(0.5) - f(1.0 / 48.0) * theta_sq + f(1.0 / 3840.0) * p4
        real = f(1.0) - f(1.0 / 8.0) * theta_sq + f(1.0 / 384.0) * p4
    else:
        theta = np.sqrt(theta_sq)
        imag = np.sin(f(0.5) * theta) / theta
        real = np.cos(f(0.5) * theta)
    q = np.array([imag * phi[0], imag * phi[1], imag * phi[2], real], dtype=f)
    theta = np.sqrt(theta_sq)
    one = f(1.0)
    if abs(sigma) < EPS:
        C = one
        if abs(theta) < EPS:
            A, B = f(0.5), f(1.0 / 6.0)
        else:
            A = (one - np.cos(theta)) / theta_sq
            B = (theta - np.sin(theta)) / (theta_sq * theta)
    else:
        C = (scale - one) / sigma
        if abs(theta) < EPS:
            s2 = sigma * sigma
            A = ((sigma - one) * scale + one) / s2
            B = (scale * f(0.5) * s2 + scale - one - sigma * scale) / (s2 * sigma)
        else:
            a, b, c = scale * np.sin(theta), scale * np.cos(theta), theta_sq + sigma * sigma
            A = (a * sigma + (one - b) * theta) / (theta * c)
            B = (C - ((b - one) * sigma + a * theta) / c) / theta_sq
    t = C * tau
    tau = np.cross(phi, tau).astype(f)
    t = t + A * tau
    tau = np.cross(phi, tau).astype(f)
    t = t + B * tau
    return t.astype(f), q, f(scale)


def retr_sim3(xi, pose):
    """pose [8] float32 (t, q, s) <- exp(xi) * pose (gn_kernels.cu:387-413)."""
    dt, dq, ds = exp_sim3(xi)
    t, q, s = pose[0:3], pose[3:7], pose[7]
    out = np.empty(8, dtype=np.float32)
    out[0:3] = act_so3(dq, t[None])[0] * ds + dt
    out[3:7] = quat_comp(dq, q)
    out[7] = ds * s
    return out


def huber(r):
    a = np.abs(r)
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(a < 1.345, 1.0, 1.345 / a)


# ---------------------------------------------------------------- per-factor blocks
def _rows(kind, P, Xi, ind, valid, q, prm):
    """-> list of (J0 [n,7], err [n], w [n]) for the residual rows of this factor kind."""
    n = P.shape[0]
    z, o = np.zeros(n), np.ones(n)
    sq = np.sqrt(np.where(valid, q, 0.0))
    if kind == "points":
        sw = sq / prm["sigma_point"]
        err = P - Xi
        x, y, zz = P[:, 0], P[:, 1], P[:, 2]
        J = [np.stack([o, z, z, z, zz, -y, x], 1), np.stack([z, o, z, -zz, z, x, y], 1), np.stack([z, z, o, y, -x, z, zz], 1)]
        return [(J[r], err[:, r], huber(sw * err[:, r]) * sw * sw) for r in range(3)]
    if kind == "rays":
        n1i = np.linalg.norm(Xi, axis=1)
        n1j = np.linalg.norm(P, axis=1)
        ri, r = Xi / n1i[:, None], P / n1j[:, None]
        err = np.concatenate([r - ri, (n1j - n1i)[:, None]], 1)
        swr, swd = sq / prm["sigma_ray"], sq / prm["sigma_dist"]
        n3 = 1.0 / (n1j ** 3)
        d = lambda a, b: (1.0 / n1j if a == b else 0.0) - P[:, a] * P[:, b] * n3
        J = [np.stack([d(0, 0), d(0, 1), d(0, 2), z, r[:, 2], -r[:, 1], z], 1),
             np.stack([d(0, 1), d(1, 1), d(1, 2), -r[:, 2], z, r[:, 0], z], 1),
             np.stack([d(0, 2), d(1, 2), d(2, 2), r[:, 1], -r[:, 0], z, z], 1),
             np.stack([r[:, 0], r[:, 1], r[:, 2], z, z, z, n1j], 1)]
        sws = [swr, swr, swr, swd]
        return [(J[k], err[:, k], huber(sws[k] * err[:, k]) * sws[k] ** 2) for k in range(4)]
    if kind == "calib":
        K, W, H = prm["K"], prm["width"], prm["height"]
        fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
        u_t, v_t = (ind % W).astype(np.float64), (ind // W).astype(np.float64)
        vz = (P[:, 2] > prm["z_eps"]) & (Xi[:, 2] > prm["z_eps"])
        with np.errstate(divide="ignore", invalid="ignore"):
            zinv = np.where(vz, 1.0 / P[:, 2], 0.0)
            zj_log = np.where(vz, np.log(np.where(vz, P[:, 2], 1.0)), 0.0)
            zi_log = np.where(vz, np.log(np.where(vz, Xi[:, 2], 1.0)), 0.0)
        xz, yz = P[:, 0] * zinv, P[:, 1] * zinv
        u, v = fx * xz + cx, fy * yz + cy
        pb = prm["pixel_border"]
        ok = valid & vz & (u > pb) & (u < W - 1 - pb) & (v > pb) & (v < H - 1 - pb)
        sq = np.sqrt(np.where(ok, q, 0.0))
        swp, swd = sq / prm["sigma_pixel"], sq / prm["sigma_depth"]
        err = np.stack([u - u_t, v - v_t, zj_log - zi_log], 1)
        J = [np.stack([fx * zinv, z, -fx * xz * zinv, -fx * xz * yz, fx * (1 + xz * xz), -fx * yz, z], 1),
             np.stack([z, fy * zinv, -fy * yz * zinv, -fy * (1 + yz * yz), fy * xz * yz, fy * xz, z], 1),
             np.stack([z, z, zinv, yz, -xz, z, o], 1)]
        sws = [swp, swp, swd]
        return [(J[k], err[:, k], huber(sws[k] * err[:, k]) * sws[k] ** 2) for k in range(3)]
    raise ValueError(kind)


def factor_blocks(kind, Twc, Xs, Cs, ii_edge, jj_edge, idx, valid_match, Q, prm):
    """Hs [4,E,7,7], gs [2,E,7] in float64, layout of gn_kernels.cu:713-745."""
    E = len(ii_edge)
    Hs, gs = np.zeros((4, E, 7, 7)), np.zeros((2, E, 7))
    T = Twc.astype(np.float64)
    for e in range(E):
        ix, jx = int(ii_edge[e]), int(jj_edge[e])
        ti, qi, si = T[ix, 0:3], T[ix, 3:7], T[ix, 7]
        tj, qj, sj = T[jx, 0:3], T[jx, 3:7], T[jx, 7]
        tij, qij, sij = rel_sim3(ti, qi, si, tj, qj, sj)
        vm = valid_match[e].reshape(-1).astype(bool)
        ind = np.where(vm, idx[e], 0)
        Xi = Xs[ix][ind].astype(np.float64)
        Xj = Xs[jx].astype(np.float64)
        P = act_so3(qij, Xj) * sij + tij
        q = Q[e].reshape(-1).astype(np.float64)
        ci, cj = Cs[ix].reshape(-1)[ind], Cs[jx].reshape(-1)
        valid = vm & (q > prm["Q_thresh"]) & (ci > prm["C_thresh"]) & (cj > prm["C_thresh"])
        H = np.zeros((14, 14))
        g = np.zeros(14)
        for J0, err, w in _rows(kind, P, Xi, ind, valid, q, prm):
            Jj = apply_sim3_adj_inv(ti, qi, si, J0)
            Jx = np.concatenate([-Jj, Jj], 1)
            H += (Jx * w[:, None]).T @ Jx
            g += (Jx * (w * err)[:, None]).sum(0)
        Hs[0, e], Hs[1, e], Hs[2, e], Hs[3, e] = H[:7, :7], H[:7, 7:], H[7:, :7], H[7:, 7:]
        gs[0, e], gs[1, e] = g[:7], g[7:]
    return Hs, gs


def solve_step(Hs, gs, ii_edge, jj_edge, num_poses, num_fix=1):
    """SparseBlock.update_lhs / update_rhs / solve (gn_kernels.cu:58-160): dx [P-fix,7] float32, zeros on failure."""
    D = 7 * (num_poses - num_fix)
    A, b = np.zeros((D, D)), np.zeros(D)
    Hs32 = Hs.astype(np.float32).astype(np.float64)  # the reference stores the blocks in float32
    gs32 = gs.astype(np.float32).astype(np.float64)
    for e in range(len(ii_edge)):
        io, jo = int(ii_edge[e]) - num_fix, int(jj_edge[e]) - num_fix
        for blk, (r, c) in enumerate(((io, io), (io, jo), (jo, io), (jo, jo))):
            if r >= 0 and c >= 0:
                A[7 * r:7 * r + 7, 7 * c:7 * c + 7] += Hs32[blk, e]
        if io >= 0:
            b[7 * io:7 * io + 7] += gs32[0, e]
        if jo >= 0:
            b[7 * jo:7 * jo + 7] += gs32[1, e]
    try:
        L = np.linalg.cholesky(A)
        x = np.linalg.solve(L.T, np.linalg.solve(L, b))
    except np.linalg.LinAlgError:
        x = np.zeros(D)
    return (-x).reshape(-1, 7).astype(np.float32)


def gauss_newton(kind, Twc, Xs, Cs, ii, jj, idx, valid_match, Q, prm, max_iter, delta_thresh, num_fix=1):
    """Full solve; Twc [P,8] float32 is updated in place like the reference; returns the last dx."""
    uniq = np.unique(np.concatenate([ii, jj]))
    ii_edge, jj_edge = np.searchsorted(uniq, ii), np.searchsorted(uniq, jj)
    P = Xs.shape[0]
    dx = np.zeros((P - num_fix, 7), dtype=np.float32)
    for _ in range(max_iter):
        Hs, gs = factor_blocks(kind, Twc, Xs, Cs, ii_edge, jj_edge, idx, valid_match, Q, prm)
        dx = solve_step(Hs, gs, ii_edge, jj_edge, P, num_fix)
        for k in range(num_fix, P):
            Twc[k] = retr_sim3(dx[k - num_fix], Twc[k])
        if np.linalg.norm(dx.astype(np.float32)) < delta_thresh:
            break
    return dx


# ---------------------------------------------------------------- residuals (for finite-difference checks)
def residual_vector(kind, Twc, Xs, ii_edge, jj_edge, idx, prm):
    """Unweighted residuals of every (factor, point), all matches valid -- used to check J0 / adjoint by differences."""
    out = []
    T = Twc.astype(np.float64)
    for e in range(len(ii_edge)):
        ix, jx = int(ii_edge[e]), int(jj_edge[e])
        tij, qij, sij = rel_sim3(T[ix, 0:3], T[ix, 3:7], T[ix, 7], T[jx, 0:3], T[jx, 3:7], T[jx, 7])
        Xi, Xj = Xs[ix][idx[e]].astype(np.float64), Xs[jx].astype(np.float64)
        P = act_so3(qij, Xj) * sij + tij
        if kind == "points":
            out.append((P - Xi).reshape(-1))
        elif kind == "rays":
            n1i, n1j = np.linalg.norm(Xi, axis=1), np.linalg.norm(P, axis=1)
            out.append(np.concatenate([P / n1j[:, None] - Xi / n1i[:, None], (n1j - n1i)[:, None]], 1).reshape(-1))
        else:
            K, W = prm["K"], prm["width"]
            u = K[0, 0] * P[:, 0] / P[:, 2] + K[0, 2]
            v = K[1, 1] * P[:, 1] / P[:, 2] + K[1, 2]
            out.append(np.stack([u - idx[e] % W, v - idx[e] // W, np.log(P[:, 2]) - np.log(Xi[:, 2])], 1).reshape(-1))
    return np.concatenate(out)

"""ORACLE (test infrastructure only): the frontend Sim(3) tracker, numpy restatement.

Follows, function by function:
  VSLAM/CameraTracker.py:53-155   track()  (masks :83-87, lost test :90-118, pose :123-135, fusion :136-141,
                                  keyframe decisions :144-153)
  :159-167 check_keyframe, :170-186 check_keyframe_map, :189-219 get_points_poses, :223-238 solve,
  :296-396 opt_pose_calib_sim3 (covariance filter :335-346; --optimize_focal :308-320,367-377: re-backprojection of the
                                  frame points with the current focal, 8th Jacobian column, K updated in place)
  VSLAM/mast3r_slam/geometry.py:38-43,116-124 constrain_points_to_ray / backproject, :47-54 act_Sim3,
  :66-113 project_calib;  nonlinear_optimizer.py:5-26 check_convergence, :29-34 huber
  VSLAM/utils_uncertainty.py:5-53 local_diag_cov_from_X1;  VSLAM/ImageFrame.py:30-52 update_pointmap / get_average_conf
The Lie algebra (`pypose`, pip dependency, README.md:72, unpinned, not in /root/reference) is restated from its
published definitions: Sim3 = (t, q xyzw, s), Act(p) = s R p + t, left-multiplicative update Exp(tau) o T.

PINNED: tests/golden/tracker_*.npz are outputs of the reference's own CameraTracker.track() executed on CPU
(tests/golden/make_golden_tracker.py; only the LieTensor algebra and the network are substituted); tests/test_tracker.py
checks this restatement against them (per-iteration tau and cost, final pose, keyframe decisions, fused pointmap).

`det_mode`: the reference takes torch.det (LU, fp32) of J S J^T; "analytic" uses det = (fx fy / Z^3)^2 s^6 vx vy vz,
which is what the HIP kernel computes (same quantity, better conditioned; differences only move points that sit
exactly on the 0.9-quantile threshold).
"""
from __future__ import annotations

import math

import numpy as np

from oracle.gn_oracle import act_so3, exp_sim3, quat_comp

F = np.float32

BASE_CFG = dict(min_match_frac=0.05, max_iters=50, C_conf=0.0, Q_conf=1.5, rel_error=1e-3, delta_norm=1e-3, huber=1.345,
                match_frac_thresh=0.333, sigma_pixel=1.0, sigma_depth=10.0, pixel_border=-10, depth_eps=1e-6)  # config/base.yaml:19-34


# ------------------------------------------------------------------------------------------------ Sim(3), pose = [8]
def quat2unit(T):
    T = T.copy()
    T[3:7] = T[3:7] / np.sqrt((T[3:7] * T[3:7]).sum(dtype=T.dtype))
    return T


def sim3_act(T, X):
    return (act_so3(T[3:7], X) * T[7] + T[0:3]).astype(X.dtype)


def sim3_inv(T):
    qi = T[3:7] * np.array([-1, -1, -1, 1], dtype=T.dtype)
    out = np.empty(8, dtype=T.dtype)
    out[0:3] = -act_so3(qi, T[None, 0:3])[0] / T[7]
    out[3:7] = qi
    out[7] = 1.0 / T[7]
    return out


def sim3_mul(A, B):
    out = np.empty(8, dtype=A.dtype)
    out[0:3] = act_so3(A[3:7], B[None, 0:3])[0] * A[7] + A[0:3]
    out[3:7] = quat_comp(A[3:7], B[3:7])
    out[7] = A[7] * B[7]
    return out


def sim3_retract(tau, T):
    """Exp(tau) o T, float32 (CameraTracker.py:373)."""
    dt, dq, ds = exp_sim3(np.asarray(tau, dtype=F))
    D = np.concatenate([dt, dq, [ds]]).astype(F)
    return sim3_mul(D, T.astype(F))


# ------------------------------------------------------------------------------------------------ geometry
def pixel_grid(H, W, dtype=F):
    u, v = np.meshgrid(np.arange(W), np.arange(H))
    return np.stack([u.reshape(-1), v.reshape(-1)], -1).astype(dtype)


def constrain_points_to_ray(X, K, H, W):
    """geometry.py:38-43 + backproject :116-124: keep z, put x, y on the pixel's ray."""
    uv = pixel_grid(H, W, X.dtype)
    z = X[:, 2]
    return np.stack([(uv[:, 0] - K[0, 2]) / K[0, 0] * z, (uv[:, 1] - K[1, 2]) / K[1, 1] * z, z], -1).astype(X.dtype)


def local_diag_var(X, H, W, win=5, var_floor=1e-12):
    """utils_uncertainty.py:5-53: per-pixel variance of x, y, z over a win x win window (reflect padding, weights =
    finite & z > 0); returns the diagonal [n,3]."""
    pad = win // 2
    Xv = X.reshape(H, W, 3)
    valid = (np.isfinite(Xv).all(-1) & (Xv[..., 2] > 0)).astype(X.dtype)[..., None]

    def box(img):
        p = np.pad(img, ((pad, pad), (pad, pad), (0, 0)), mode="reflect")
        acc = np.zeros_like(img)
        for dy in range(win):
            for dx in range(win):
                acc = acc + p[dy:dy + H, dx:dx + W]
        return acc / X.dtype.type(win * win)

    denom = np.maximum(box(valid), X.dtype.type(1e-9))
    mean = box(Xv * valid) / denom
    ex2 = box(Xv * Xv * valid) / denom
    return np.maximum(ex2 - mean * mean, X.dtype.type(var_floor)).reshape(-1, 3)


def project_calib(P, K, H, W, border, z_eps, dP_df=None):
    """geometry.py:66-113: pz [n,3] = (u, v, log z), dpz_dP [n,3,3], valid [n]; with dP_df [n,3] (= dXf_Ck_d_f) also the
    focal column [n,3] exactly as the reference writes it (:110-112, including its division by z_inv**2)."""
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    x, y, z = P[:, 0], P[:, 1], P[:, 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        u = (fx * x + cx * z) / z   # p = K P; p / p[2]
        v = (fy * y + cy * z) / z
        valid_z = z > z_eps
        valid = (u > border) & (u < W - 1 - border) & (v > border) & (v < H - 1 - border) & valid_z
        logz = np.where(valid_z, np.log(np.where(valid_z, z, 1)), 0).astype(P.dtype)
        z_inv = 1.0 / z
    J = np.zeros((len(P), 3, 3), dtype=P.dtype)
    J[:, 0, 0] = fx * z_inv
    J[:, 1, 1] = fy * z_inv
    J[:, 0, 2] = -fx * x * z_inv * z_inv
    J[:, 1, 2] = -fy * y * z_inv * z_inv
    J[:, 2, 2] = z_inv
    if dP_df is not None:
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            d0, d1, d2 = dP_df[:, 0], dP_df[:, 1], dP_df[:, 2]
            zi2 = z_inv * z_inv                                   # z_inv ** 2
            Jf = np.stack([x * z_inv + fx * (d0 * z - d2 * x) / zi2, y * z_inv + fy * (d1 * z - d2 * y) / zi2, z_inv * d2], -1).astype(P.dtype)
        return np.stack([u, v, logz], -1).astype(P.dtype), J, valid, Jf
    return np.stack([u, v, logz], -1).astype(P.dtype), J, valid


def act_jacobian(P):
    """geometry.py:47-54: d(T p)/d tau for the left perturbation = [I, -skew(pW), pW]  [n,3,7]."""
    n = len(P)
    J = np.zeros((n, 3, 7), dtype=P.dtype)
    J[:, 0, 0] = J[:, 1, 1] = J[:, 2, 2] = 1
    x, y, z = P[:, 0], P[:, 1], P[:, 2]
    J[:, 0, 4], J[:, 0, 5] = z, -y
    J[:, 1, 3], J[:, 1, 5] = -z, x
    J[:, 2, 3], J[:, 2, 4] = y, -x
    J[:, :, 6] = P
    return J


def rot_scale_matrix(T):
    """T.matrix()[:3,:3] = s R."""
    return (act_so3(T[3:7], np.eye(3, dtype=T.dtype)).T * T[7]).astype(T.dtype)


def pixel_cov_det(P, T, var, K, det_mode):
    """CameraTracker.py:335-344."""
    fx, fy = K[0, 0], K[1, 1]
    X, Y, Z = P[:, 0], P[:, 1], P[:, 2]
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        if det_mode == "analytic":
            s3 = T[7] * T[7] * T[7]
            jd = fx * fy / (Z * Z * Z) * s3
            return (jd * jd * (var[:, 0] * var[:, 1] * var[:, 2])).astype(P.dtype)
        sR = rot_scale_matrix(T)
        cov = np.einsum("ab,nb,cb->nac", sR, var, sR).astype(P.dtype)
        o = np.zeros_like(X)
        JC = np.stack([fx / Z, o, -fx * X / Z ** 2, o, fy / Z, -fy * Y / Z ** 2, o, o, 1 / Z], -1).reshape(-1, 3, 3).astype(P.dtype)
        M = (JC @ cov @ JC.transpose(0, 2, 1)).astype(P.dtype)
        return np.linalg.det(M).astype(P.dtype)


def quantile_linear(x, q):
    """torch.quantile(x, q) for a 1-D float32 tensor: rank = float32(q) * (n - 1) in float32, linear interpolation with
    torch.lerp's two-sided form.  NaNs are not handled (torch returns NaN)."""
    x = np.sort(np.asarray(x, dtype=F))
    n = len(x)
    rank = F(F(q) * F(n - 1))
    lo = int(np.floor(rank))
    hi = int(np.ceil(rank))
    w = F(rank - F(lo))
    a, b = x[lo], x[hi]
    with np.errstate(invalid="ignore", over="ignore"):
        return F(a + w * (b - a)) if w < F(0.5) else F(b - (b - a) * (F(1) - w))


def huber(r, k):
    a = np.abs(r)
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(a < k, 1.0, k / a).astype(r.dtype)


def check_convergence(rel_thr, delta_thr, old_cost, new_cost, delta):
    """nonlinear_optimizer.py:5-26 (python floats: old_cost = inf gives nan, which compares False)."""
    cost_diff = old_cost - new_cost
    rel_dec = math.fabs(cost_diff / old_cost) if old_cost != 0 else float("nan")
    return bool(rel_dec < rel_thr or float(np.linalg.norm(delta)) < delta_thr)


# ------------------------------------------------------------------------------------------------ the optimisation
def normal_equations(T, Xf, var_f, meas_k, valid_meas_k, sqrt_info, K, H, W, cfg, covariance_filter, det_mode, dXf_df=None):
    """One linearisation (CameraTracker.py:321-372 + solve :223-234): H [7,7], g [7], cost, threshold used.  With dXf_df [n,3]
    (optimize_focal: derivative of the re-backprojected frame point w.r.t. the focal, :313-316) the system is 8 x 8."""
    dt = Xf.dtype
    P = sim3_act(T.astype(dt), Xf)
    Jf = None
    if dXf_df is not None:
        dP_df = (dXf_df.astype(dt) @ rot_scale_matrix(T.astype(dt)).T).astype(dt)     # T.matrix()[:3,:3] @ dXf_f  (:323,333)
        pz, dpz, valid_proj, Jf = project_calib(P, K.astype(dt), H, W, cfg["pixel_border"], cfg["depth_eps"], dP_df)
    else:
        pz, dpz, valid_proj = project_calib(P, K.astype(dt), H, W, cfg["pixel_border"], cfg["depth_eps"])
    thr = None
    if covariance_filter:
        det = pixel_cov_det(P, T.astype(dt), var_f, K.astype(dt), det_mode)
        thr = max(float(quantile_linear(det, 0.9)), 1.0)
        valid_cov = det < thr
    else:
        valid_cov = np.ones(len(P), bool)
    valid2 = valid_proj & valid_meas_k & valid_cov
    si2 = sqrt_info * valid2[:, None].astype(dt)
    r = meas_k - pz
    J = -(dpz @ act_jacobian(P))
    if Jf is not None:
        J = np.concatenate([J, -Jf[..., None]], -1)                                     # :368-369
    whitened = si2 * r
    robust = si2 * np.sqrt(huber(whitened, dt.type(cfg["huber"])))
    A = (robust[..., None] * J).reshape(-1, J.shape[-1])
    b = (robust * r).reshape(-1, 1)
    Hm = A.T @ A
    g = -(A.T @ b)[:, 0]
    cost = 0.5 * float((b.T @ b)[0, 0])
    return Hm, g, cost, thr


def opt_pose_calib_sim3(Xf, var_f, T_WCf, T_WCk, Qk, valid, meas_k, valid_meas_k, K, H, W, cfg, covariance_filter=True,
                        det_mode="lu", dtype=F, trace=None, idx_f2k=None):
    """CameraTracker.py:296-396.  Xf / var_f / Qk / valid / meas_k / valid_meas_k are in keyframe pixel order.
    Returns (T_WCf [8], T_CkCf [8], iterations); raises np.linalg.LinAlgError where torch.linalg.cholesky would.
    idx_f2k given = --optimize_focal: K [3,3] is UPDATED IN PLACE (the reference writes self.K_slam, :376-377)."""
    Xf, var_f = Xf.astype(dtype), (var_f.astype(dtype) if var_f is not None else None)
    focal = idx_f2k is not None
    if focal:
        uf, vf = (idx_f2k % W).astype(dtype), (idx_f2k // W).astype(dtype)
    w = (valid[:, 0] * np.sqrt(Qk[:, 0])).astype(dtype)
    sqrt_info = np.stack([w / dtype(cfg["sigma_pixel"])] * 2 + [w / dtype(cfg["sigma_depth"])], -1).astype(dtype)
    T = sim3_mul(sim3_inv(T_WCk.astype(dtype)), T_WCf.astype(dtype))
    old_cost = float("inf")
    it = 0
    for step in range(int(cfg["max_iters"])):
        dXf_df = None
        if focal:                                                                       # :308-318
            z = Xf[:, 2]
            dXf_df = np.stack([-(uf - K[0, 2]) / (K[0, 0] ** 2) * z, -(vf - K[1, 2]) / (K[1, 1] ** 2) * z, np.zeros_like(z)], -1).astype(dtype)
            Xf = np.stack([(uf - K[0, 2]) / K[0, 0] * z, (vf - K[1, 2]) / K[1, 1] * z, z], -1).astype(dtype)
        Hm, g, cost, thr = normal_equations(T, Xf, var_f, meas_k.astype(dtype), valid_meas_k, sqrt_info, K, H, W, cfg,
                                            covariance_filter, det_mode, dXf_df)
        L = np.linalg.cholesky(Hm.astype(dtype))
        if not np.isfinite(L).all():
            raise np.linalg.LinAlgError("cholesky")
        y = np.linalg.solve(L, g.astype(dtype))
        tau = np.linalg.solve(L.T, y).astype(dtype)
        if trace is not None:
            trace.append(dict(H=Hm, g=g, cost=cost, tau=tau, thr=thr, T=T.copy(), fx=float(K[0, 0])))
        T = quat2unit(sim3_retract(tau[:7], T).astype(dtype))
        if focal:                                                                       # :376-377
            K[0, 0] = K[0, 0] + tau[7]
            K[1, 1] = K[1, 1] + tau[7]
        it = step + 1
        if check_convergence(cfg["rel_error"], cfg["delta_norm"], old_cost, cost, tau[:7]):
            break
        old_cost = cost
    return sim3_mul(T_WCk.astype(dtype), T), T, it


# ------------------------------------------------------------------------------------------------ track()
def update_pointmap(X_canon, C, N, X, Cn):
    """ImageFrame.update_pointmap (ImageFrame.py:30-48): confidence-weighted running mean."""
    if N == 0:
        return X.copy(), Cn.copy(), 1
    return ((C * X_canon + Cn * X) / (C + Cn)).astype(X.dtype), (C + Cn).astype(X.dtype), N + 1


def track(sc, cfg=None, covariance_filter=True, min_displacement=30.0, thres_keyframe=0.8, last_dist=0.0, det_mode="lu",
          kf_N=None, trace=None, optimize_focal=False, K=None):
    """CameraTracker.track (:53-155) on a `tracker_scene` dict for a FRESH frame (frame.N == 0) and a keyframe that holds
    (Xk_canon, Ck, kf_N).  Returns a dict with the reference's observable effects."""
    cfg = dict(BASE_CFG, **(cfg or {}))
    kf_N = int(sc.get("kf_N", 1)) if kf_N is None else kf_N
    H, W = sc["height"], sc["width"]
    K = (sc["K"] if K is None else K).astype(F).copy()     # optimize_focal updates it; returned as out["K"]
    n = H * W
    idx = sc["idx_f2k"]
    vm = sc["valid_match"][:, 0]
    Qk = np.sqrt(sc["Qff"][idx] * sc["Qkf"])
    Xf_canon, Cf_tot, Nf = update_pointmap(None, None, 0, sc["Xff"], sc["Cff"])   # :74
    T_WCf, T_WCk = quat2unit(sc["T_WCf0"][0].astype(F)), quat2unit(sc["T_WCk"][0].astype(F))
    Cf, Ck = Cf_tot / Nf, sc["Ck"] / kf_N
    Xf_c = constrain_points_to_ray(Xf_canon, K, H, W)
    Xk_c = constrain_points_to_ray(sc["Xk_canon"], K, H, W)
    var_f = local_diag_var(Xf_c, H, W)
    with np.errstate(divide="ignore", invalid="ignore"):
        logz = np.log(Xk_c[:, 2:3])
    meas_k = np.concatenate([pixel_grid(H, W), logz], -1).astype(F)
    valid_meas_k = Xk_c[:, 2] > cfg["depth_eps"]
    meas_k[~valid_meas_k] = 0.0
    valid_opt = vm & (Cf[idx, 0] > cfg["C_conf"]) & (Ck[:, 0] > cfg["C_conf"]) & (Qk[:, 0] > cfg["Q_conf"])
    valid_kf = vm & (Qk[:, 0] > cfg["Q_conf"])
    out = dict(lost=False, is_keyframe=False, is_keyframe_map=False, T_WCf=sc["T_WCf0"][0].copy(), iterations=0,
               last_dist=last_dist, kf_X=sc["Xk_canon"], kf_C=sc["Ck"], kf_N=kf_N, n_opt=int(valid_opt.sum()),
               Xf_c=Xf_c, var_f=var_f, valid_opt=valid_opt, valid_kf=valid_kf, Qk=Qk, meas_k=meas_k, valid_meas_k=valid_meas_k, K=K)
    if F(valid_opt.sum()) / F(n) < cfg["min_match_frac"]:
        out["lost"] = True
        return out
    try:
        T_new, T_CkCf, its = opt_pose_calib_sim3(Xf_c[idx], var_f[idx], T_WCf, T_WCk, Qk, valid_opt[:, None], meas_k, valid_meas_k,
                                                 K, H, W, cfg, covariance_filter, det_mode, trace=trace,
                                                 idx_f2k=idx if optimize_focal else None)
    except np.linalg.LinAlgError:
        out["lost"] = True
        return out
    out["T_WCf"], out["T_CkCf"], out["iterations"] = quat2unit(T_new), T_CkCf, its
    Xkk = sim3_act(T_CkCf, sc["Xkf"].astype(F))                                           # :138
    out["kf_X"], out["kf_C"], out["kf_N"] = update_pointmap(sc["Xk_canon"], sc["Ck"], kf_N, Xkk, sc["Ckf"])
    # check_keyframe (:159-167)
    match_frac_k = F(valid_kf.sum()) / F(n)
    unique_frac_f = len(np.unique(idx[vm])) / n
    out["n_kf"], out["n_unique"] = int(valid_kf.sum()), len(np.unique(idx[vm]))
    out["is_keyframe"] = bool(min(float(match_frac_k), unique_frac_f) < cfg["match_frac_thresh"])
    if out["is_keyframe"]:
        out["is_keyframe_map"], out["last_dist"] = True, 0
    else:                                                                                 # check_keyframe_map (:170-186)
        uvf = np.stack([idx % W, idx // W], -1)
        uvk = pixel_grid(H, W, np.int64)
        d = np.sqrt(((uvf - uvk).astype(F) ** 2).sum(-1, dtype=F))[valid_opt]
        dq = float(quantile_linear(d, thres_keyframe))
        out["dist_quantile"] = dq
        out["is_keyframe_map"] = bool((dq - last_dist) > min_displacement)
        if out["is_keyframe_map"]:
            out["last_dist"] = dq
    return out

"""Seeded synthetic workloads for the benches and tests (numpy, host side; no kernel code, no oracle code).

`keyframe_graph` / `calib_keyframe_graph` build the inputs of the Sim(3) global optimiser
(`mast3r_slam_backends.gauss_newton_*`, VSLAM/mast3r_slam/global_opt.py:138-231): per-keyframe canonical pointmaps
`Xs [P,n,3]`, confidences `Cs [P,n,1]`, a two-way factor list `ii/jj [E]` (prep_two_way_edges, global_opt.py:131-138),
matches `idx [E,n]`, `valid [E,n,1]`, match scores `Q [E,n,1]`, and the ground-truth poses `T_gt [P,8]` = (t, q xyzw, s).
The Gaussian-cloud workload of the mapper lives in `harness.mapper.synthetic_cloud`.
"""
from __future__ import annotations

import numpy as np

IDENTITY_POSE = np.array([0, 0, 0, 0, 0, 0, 1, 1], dtype=np.float32)


def quat_rotate(q, X):
    """Rotate X [...,3] by the unit quaternion q (xyzw)."""
    qv = np.broadcast_to(np.asarray(q, dtype=np.float64)[:3], X.shape)
    uv = 2.0 * np.cross(qv, X)
    return X + q[3] * uv + np.cross(qv, uv)


def quat_mul(a, b):
    return np.array([a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1],
                     a[3] * b[1] - a[0] * b[2] + a[1] * b[3] + a[2] * b[0],
                     a[3] * b[2] + a[0] * b[1] - a[1] * b[0] + a[2] * b[3],
                     a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2]])


def random_poses(rng, n, t_scale=0.3, r_scale=0.15, s_scale=0.08):
    """n Sim(3) poses near the identity: t ~ N(0, t_scale), rotation by a random axis-angle of size ~r_scale,
    scale exp(N(0, s_scale)).  Pose 0 is the identity (the optimiser holds it fixed)."""
    T = np.zeros((n, 8), dtype=np.float32)
    for k in range(n):
        w = r_scale * rng.standard_normal(3)
        th = np.linalg.norm(w)
        q = np.array([0, 0, 0, 1.0]) if th < 1e-12 else np.concatenate([np.sin(th / 2) * w / th, [np.cos(th / 2)]])
        T[k, 0:3], T[k, 3:7], T[k, 7] = t_scale * rng.standard_normal(3), q, np.exp(s_scale * rng.standard_normal())
    T[0] = IDENTITY_POSE
    return T


def perturb_poses(T, rng, mag):
    """Left-multiply every pose but the first by a small random Sim(3) (what a drifted estimate looks like)."""
    out = T.copy()
    for k in range(1, len(T)):
        d = random_poses(rng, 2, mag, mag, mag)[1].astype(np.float64)
        t, q, s = T[k, 0:3].astype(np.float64), T[k, 3:7].astype(np.float64), float(T[k, 7])
        out[k, 0:3] = quat_rotate(d[3:7], t[None])[0] * d[7] + d[0:3]
        out[k, 3:7] = quat_mul(d[3:7], q)
        out[k, 7] = d[7] * s
    return out


def relative_pose(Ti, Tj):
    """T_i^-1 T_j as (t, q, s), float64."""
    ti, qi, si = Ti[0:3].astype(np.float64), Ti[3:7].astype(np.float64), float(Ti[7])
    tj, qj, sj = Tj[0:3].astype(np.float64), Tj[3:7].astype(np.float64), float(Tj[7])
    qi_inv = np.array([-qi[0], -qi[1], -qi[2], qi[3]])
    return quat_rotate(qi_inv, (tj - ti)[None])[0] / si, quat_mul(qi_inv, qj), sj / si


def two_way_edges(rng, num_poses, extra_edges):
    """Chain 0-1-2-... plus `extra_edges` random loop closures, each factor in both directions."""
    edges = [(p, p + 1) for p in range(num_poses - 1)]
    extra_edges = min(extra_edges, num_poses * (num_poses - 1) // 2 - (num_poses - 1))  # only so many pairs exist
    while len(edges) < num_poses - 1 + extra_edges:
        a, b = rng.integers(0, num_poses, 2)
        if a != b and (a, b) not in edges and (b, a) not in edges:
            edges.append((int(a), int(b)))
    ii = np.array([a for a, b in edges] + [b for a, b in edges], dtype=np.int64)
    jj = np.array([b for a, b in edges] + [a for a, b in edges], dtype=np.int64)
    return ii, jj


def keyframe_graph(num_poses=5, n=768, seed=0, extra_edges=3, noise=0.0, outlier_frac=0.0, kf_ids=None, coherent=False):
    """EXACTLY consistent graph for the point / ray factors: n world points, keyframe p stores T_p^-1 W in a private
    slot order, a match pairs the two slots of the same world point, so with noise = 0 the ground-truth poses give
    zero residual on every valid match.  10 % of the matches get Q below the 1.5 threshold and 10 % are flagged
    invalid (with garbage indices): both must be ignored.  kf_ids: optional non-contiguous keyframe ids.
    coherent: slot orders are circular shifts instead of random permutations, so neighbouring points match
    neighbouring points as dense pixel matches do (the random order is the worst case for the gathers)."""
    rng = np.random.default_rng(seed)
    T_gt = random_poses(rng, num_poses)
    W = np.concatenate([rng.uniform(-1.5, 1.5, (n, 2)), rng.uniform(2.0, 5.0, (n, 1))], 1)
    if coherent:
        perm = [np.roll(np.arange(n), int(rng.integers(0, n))) for _ in range(num_poses)]
    else:
        perm = [rng.permutation(n) for _ in range(num_poses)]  # slot of world point m in keyframe p
    Xs = np.zeros((num_poses, n, 3), dtype=np.float32)
    for p in range(num_poses):
        t, q, s = T_gt[p, 0:3].astype(np.float64), T_gt[p, 3:7].astype(np.float64), float(T_gt[p, 7])
        Xs[p][perm[p]] = (quat_rotate(np.array([-q[0], -q[1], -q[2], q[3]]), W - t) / s).astype(np.float32)
    if noise > 0:
        Xs = (Xs + noise * rng.standard_normal(Xs.shape)).astype(np.float32)
    Cs = (1.0 + rng.random((num_poses, n, 1))).astype(np.float32)
    ii, jj = two_way_edges(rng, num_poses, extra_edges)
    E = len(ii)
    idx = np.zeros((E, n), dtype=np.int64)
    valid = np.ones((E, n, 1), dtype=bool)
    Q = (1.6 + rng.random((E, n, 1))).astype(np.float32)
    for e in range(E):
        inv_j = np.empty(n, dtype=np.int64)
        inv_j[perm[jj[e]]] = np.arange(n)
        idx[e] = perm[ii[e]][inv_j]
        Q[e, rng.random(n) < 0.1, 0] = 1.0
        bad = rng.random(n) < 0.1
        valid[e, bad, 0] = False
        idx[e][bad] = rng.integers(0, n, bad.sum())
        if outlier_frac > 0:
            out = (rng.random(n) < outlier_frac) & ~bad
            idx[e][out] = rng.integers(0, n, out.sum())
    if kf_ids is not None:
        kf_ids = np.asarray(kf_ids, dtype=np.int64)
        ii, jj = kf_ids[ii], kf_ids[jj]
    return dict(T_gt=T_gt, Xs=Xs, Cs=Cs, ii=ii, jj=jj, idx=idx, valid=valid, Q=Q)


def calib_keyframe_graph(num_poses=4, height=48, width=64, seed=0, extra_edges=2, fx=70.0):
    """Pixel-grid graph for the calibrated factor: keyframe p stores points on its own pixel rays (what
    constrain_points_to_ray produces, mast3r_slam/geometry.py:38-43) of one smooth world surface; a match pairs a pixel
    of j with the NEAREST pixel of its re-projection into i (so ground truth leaves a sub-pixel residual)."""
    rng = np.random.default_rng(seed)
    n = height * width
    K = np.array([[fx, 0, width / 2.0], [0, fx, height / 2.0], [0, 0, 1]], dtype=np.float32)
    T_gt = random_poses(rng, num_poses, 0.25, 0.05, 0.03)
    uu, vv = np.meshgrid(np.arange(width), np.arange(height))
    uv = np.stack([uu.reshape(-1), vv.reshape(-1)], 1).astype(np.float64)
    surf = lambda x, y: 3.0 + 0.3 * np.sin(1.3 * x) + 0.25 * np.cos(1.1 * y)   # world surface z = surf(x, y)
    Xs = np.zeros((num_poses, n, 3), dtype=np.float32)
    for p in range(num_poses):
        t, q, s = T_gt[p, 0:3].astype(np.float64), T_gt[p, 3:7].astype(np.float64), float(T_gt[p, 7])
        d = np.stack([(uv[:, 0] - K[0, 2]) / fx, (uv[:, 1] - K[1, 2]) / fx, np.ones(n)], 1)
        z = np.full(n, 3.0)
        for _ in range(30):                         # fixed point: camera depth whose world point lies on the surface
            Wp = quat_rotate(q, d * z[:, None]) * s + t
            z = z + (surf(Wp[:, 0], Wp[:, 1]) - Wp[:, 2]) / s
        Xs[p] = (d * z[:, None]).astype(np.float32)
    Cs = (1.0 + rng.random((num_poses, n, 1))).astype(np.float32)
    ii, jj = two_way_edges(rng, num_poses, extra_edges)
    E = len(ii)
    idx = np.zeros((E, n), dtype=np.int64)
    valid = np.zeros((E, n, 1), dtype=bool)
    Q = (1.6 + rng.random((E, n, 1))).astype(np.float32)
    for e in range(E):
        tij, qij, sij = relative_pose(T_gt[ii[e]], T_gt[jj[e]])
        P = quat_rotate(qij, Xs[jj[e]].astype(np.float64)) * sij + tij
        u = np.rint(fx * P[:, 0] / P[:, 2] + K[0, 2]).astype(np.int64)
        v = np.rint(fx * P[:, 1] / P[:, 2] + K[1, 2]).astype(np.int64)
        ok = (P[:, 2] > 0.1) & (u >= 0) & (u < width) & (v >= 0) & (v < height)
        idx[e] = np.where(ok, v * width + u, 0)
        valid[e, :, 0] = ok
        Q[e, rng.random(n) < 0.1, 0] = 1.0
    return dict(T_gt=T_gt, Xs=Xs, Cs=Cs, K=K, ii=ii, jj=jj, idx=idx, valid=valid, Q=Q, height=height, width=width)


def tracker_scene(height=48, width=64, seed=0, fx=70.0, pose_noise=0.03, depth_noise=0.01, outlier_frac=0.03,
                  kf_pose=None, drop_frac=0.0, rough_cols=0.0, kf_N=1, bad_depth_frac=0.0):
    """One frame-to-keyframe tracking problem (VSLAM/CameraTracker.py:53-155): a keyframe (pose 0 of a 2-pose
    `calib_keyframe_graph`, i.e. points on its own pixel rays of one smooth world surface) and a frame (pose 1), with
    what `mast3r_match_asymmetric` hands the tracker (utils_mast3r.py:144-170):

      Xff / Cff / Qff [n,3],[n,1],[n,1]   the frame's pointmap, confidence and descriptor confidence in its own camera
      Xkf / Ckf / Qkf                     the KEYFRAME's points as predicted in the FRAME's camera
      idx_f2k [n] int64, valid_match [n,1] bool   for every keyframe pixel, the matched frame pixel

    plus the keyframe's stored canonical pointmap `Xk_canon`/`Ck` (N = 1), the drifted initial frame pose `T_WCf0`,
    the keyframe pose `T_WCk`, the exact frame pose `T_WCf_gt` and `K`.  Depth noise makes the 5x5 local variances
    non-trivial (covariance filter); `outlier_frac` of the matches point at random frame pixels; 10 % carry a
    descriptor confidence below the 1.5 threshold; `drop_frac` of the matches are flagged invalid (a frame that
    barely overlaps the keyframe); the right-hand `rough_cols` of the frame's columns get a very noisy depth
    (local pixel-covariance determinants above 1: the 0.9-quantile branch of the covariance filter).  `kf_N` > 1: the
    keyframe's stored map is the confidence-weighted mean of kf_N predictions (`Ck` is their SUM, ImageFrame.py:30-52);
    `bad_depth_frac` of the pixels of both maps get a negative depth (behind the camera: invalid measurements).  float32 / int64 / bool numpy arrays."""
    rng = np.random.default_rng(seed + 1000)
    g = calib_keyframe_graph(num_poses=2, height=height, width=width, seed=seed, extra_edges=0, fx=fx)
    n = height * width
    T = g["T_gt"].copy()
    if kf_pose is not None:  # move both poses by a common world transform so that the keyframe is not the identity
        for k in range(2):
            t, q, s = T[k, 0:3].astype(np.float64), T[k, 3:7].astype(np.float64), float(T[k, 7])
            T[k, 0:3] = quat_rotate(kf_pose[3:7].astype(np.float64), t[None])[0] * kf_pose[7] + kf_pose[0:3]
            T[k, 3:7] = quat_mul(kf_pose[3:7].astype(np.float64), q)
            T[k, 7] = kf_pose[7] * s
    e = int(np.nonzero((g["ii"] == 1) & (g["jj"] == 0))[0][0])  # points of the keyframe (j = 0) matched into the frame (i = 1)
    idx, valid = g["idx"][e].copy(), g["valid"][e].copy()
    out = (rng.random(n) < outlier_frac) & valid[:, 0]
    idx[out] = rng.integers(0, n, out.sum())
    if drop_frac > 0:
        valid[rng.random(n) < drop_frac, 0] = False
    noisy = lambda X: (X * (1.0 + depth_noise * rng.standard_normal((n, 1))) + 0.002 * rng.standard_normal((n, 3))).astype(np.float32)
    Xk_canon, Xff = noisy(g["Xs"][0]), noisy(g["Xs"][1])
    if rough_cols > 0:
        rough = (np.arange(n) % width) >= int(round(width * (1.0 - rough_cols)))
        Xff[rough] *= np.exp(0.6 * rng.standard_normal((int(rough.sum()), 1))).astype(np.float32)
    # the keyframe's points seen from the frame: T_CfCk Xk
    tfk, qfk, sfk = relative_pose(g["T_gt"][1], g["T_gt"][0])
    Xkf = noisy(quat_rotate(qfk, g["Xs"][0].astype(np.float64)) * sfk + tfk)
    conf = lambda lo: (lo + rng.random((n, 1))).astype(np.float32)
    Qff, Qkf = conf(1.6), conf(1.6)
    Qkf[rng.random(n) < 0.1, 0] = 1.0
    T0 = perturb_poses(T, rng, pose_noise)
    Ck = conf(1.0)
    for _ in range(int(kf_N) - 1):  # further predictions fused into the keyframe (update_pointmap)
        Xn, Cn = noisy(g["Xs"][0]), conf(1.0)
        Xk_canon = ((Ck * Xk_canon + Cn * Xn) / (Ck + Cn)).astype(np.float32)
        Ck = (Ck + Cn).astype(np.float32)
    if bad_depth_frac > 0:
        for X in (Xk_canon, Xff):
            bad = rng.random(n) < bad_depth_frac
            X[bad, 2] = -np.abs(X[bad, 2])
    return dict(kf_N=int(kf_N), height=height, width=width, K=g["K"], Xff=Xff, Cff=conf(1.0), Qff=Qff, Xkf=Xkf, Ckf=conf(1.0), Qkf=Qkf,
                Xk_canon=Xk_canon, Ck=Ck, idx_f2k=idx, valid_match=valid, T_WCk=T[0:1].copy(), T_WCf0=T0[1:2].copy(),
                T_WCf_gt=T[1:2].copy())

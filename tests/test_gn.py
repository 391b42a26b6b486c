"""Sim(3) Gauss-Newton global optimiser (mast3r_slam_backends.gauss_newton_points / rays / calib).

CPU: the oracle's Jacobians against finite differences of its own residuals, and recovery of known poses.
GPU: the HIP path (through the drop-in, i.e. through the C ABI) against the oracle -- per-factor blocks, full solves --
plus size-independent properties at the reference's full factor size (196 608 points per factor)."""
import numpy as np
import pytest
import torch

from artdeco_amd import synthetic as S
from oracle import gn_oracle as G

PRM = dict(sigma_point=0.05, sigma_ray=0.003, sigma_dist=10.0, C_thresh=0.0, Q_thresh=1.5)  # config/base.yaml:36-51
KINDS = ("points", "rays", "calib")


def _calib_prm(c):
    return dict(PRM, K=c["K"], width=c["width"], height=c["height"], pixel_border=-10, z_eps=1e-6, sigma_pixel=1.0, sigma_depth=10.0)


def _perturb(T_gt, seed, mag):
    return S.perturb_poses(T_gt, np.random.default_rng(seed), mag)


def _graph(kind, **kw):
    if kind == "calib":
        c = S.calib_keyframe_graph(**kw)
        return c, _calib_prm(c)
    return S.keyframe_graph(**kw), dict(PRM)


# ------------------------------------------------------------------------------------------ CPU: oracle vs the reference's Python
@pytest.mark.parametrize("kind", ["rays", "calib"])
def test_oracle_single_factor_matches_reference_tracker_code(kind):
    """tests/golden/gn_factor_*.npz: the (i, j) factor of a two-keyframe graph solved by the REFERENCE's own
    CameraTracker.opt_pose_ray_dist_sim3 / opt_pose_calib_sim3 (same residuals, weights, Huber kernel and retraction as
    gn_kernels.cu; tests/golden/make_golden_gn_factor.py).  One Gauss-Newton step of the oracle on that single factor must
    be the reference's step, iteration after iteration."""
    import os
    from conftest import GOLDEN
    d = np.load(os.path.join(GOLDEN, f"gn_factor_{kind}.npz"))
    scenes = {"rays": dict(num_poses=2, n=1500, seed=21, extra_edges=0, noise=0.003),
              "calib": dict(num_poses=2, height=40, width=56, seed=22, extra_edges=0, fx=60.0)}
    c, prm = _graph(kind, **scenes[kind])
    assert float(d["in_sum_Xs"]) == float(c["Xs"].astype(np.float64).sum())
    e = int(d["edge"])
    assert (c["ii"][e], c["jj"][e]) == (0, 1)
    T = d["T0"].astype(np.float32).copy()
    ie, je = np.array([0]), np.array([1])
    for it in range(len(d["out_taus"])):
        Hs, gs = G.factor_blocks(kind, T, c["Xs"], c["Cs"], ie, je, c["idx"][e:e + 1], c["valid"][e:e + 1], c["Q"][e:e + 1], prm)
        dx = G.solve_step(Hs, gs, ie, je, 2)
        assert np.abs(dx[0] - d["out_taus"][it]).max() < 2e-4 * max(1.0, np.abs(d["out_taus"][it]).max()), it
        cost = 0.5 * sum(float((w * err * err).sum()) for _, err, w in _factor_rows(kind, T, c, e, prm))
        assert abs(cost / d["out_costs"][it] - 1) < 2e-4, it
        T[1] = G.retr_sim3(dx[0], T[1])
    Tn = T[1].copy()
    Tn[3:7] /= np.linalg.norm(Tn[3:7])
    assert np.abs(Tn - d["out_T_j"][0]).max() < 2e-5


def _factor_rows(kind, T, c, e, prm):
    T = T.astype(np.float64)
    tij, qij, sij = G.rel_sim3(T[0, 0:3], T[0, 3:7], T[0, 7], T[1, 0:3], T[1, 3:7], T[1, 7])
    vm = c["valid"][e].reshape(-1).astype(bool)
    ind = np.where(vm, c["idx"][e], 0)
    P = G.act_so3(qij, c["Xs"][1].astype(np.float64)) * sij + tij
    q = c["Q"][e].reshape(-1).astype(np.float64)
    valid = vm & (q > prm["Q_thresh"]) & (c["Cs"][0].reshape(-1)[ind] > prm["C_thresh"]) & (c["Cs"][1].reshape(-1) > prm["C_thresh"])
    return G._rows(kind, P, c["Xs"][0][ind].astype(np.float64), ind, valid, q, prm)


# ------------------------------------------------------------------------------------------ pinned to the reference's own gn_kernels.cu
def _ref_cases():
    import os
    import sys
    from conftest import GOLDEN
    sys.path.insert(0, GOLDEN)
    import make_golden_gn_ref as M
    return M, np.load(os.path.join(GOLDEN, "ref_gn.npz"))


def _ref_prm(kind, g):
    return _calib_prm(g) if kind == "calib" else dict(PRM)


REF_CASES = ("points_a", "points_b", "rays_a", "rays_b", "calib_a")


@pytest.mark.parametrize("name", REF_CASES)
def test_oracle_matches_the_references_own_solver_on_multi_factor_graphs(name):
    """tests/golden/ref_gn.npz = outputs of the REFERENCE's gn_kernels.cu compiled for the host (oracle/_ref/ref_gn.so,
    make_golden_gn_ref.py): multi-factor graphs, all three factor kinds incl. `points`, outliers, non-contiguous keyframe ids.
    The first Gauss-Newton step (every factor's blocks, their assembly, the fixed first pose, the solve, the retraction) and
    the poses after the 10-iteration call of global_opt.py."""
    M, d = _ref_cases()
    kind, g, T0 = M.cases()[name]
    assert float(d[name + "_in_sum"]) == float(g["Xs"].astype(np.float64).sum() + T0.astype(np.float64).sum())
    prm = _ref_prm(kind, g)
    T = T0.astype(np.float32).copy()
    dx = G.gauss_newton(kind, T, g["Xs"], g["Cs"], g["ii"], g["jj"], g["idx"], g["valid"], g["Q"], prm, 1, 1e-8)
    assert np.abs(dx - d[name + "_dx1"]).max() <= 2e-4 * np.abs(d[name + "_dx1"]).max(), np.abs(dx - d[name + "_dx1"]).max()
    assert np.abs(T - d[name + "_T1"]).max() <= 2e-5
    T = T0.astype(np.float32).copy()
    dx = G.gauss_newton(kind, T, g["Xs"], g["Cs"], g["ii"], g["jj"], g["idx"], g["valid"], g["Q"], prm, 10, 1e-8)
    assert np.abs(T - d[name + "_T10"]).max() <= 2e-5, np.abs(T - d[name + "_T10"]).max()
    assert np.abs(dx - d[name + "_dx10"]).max() <= 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("name", REF_CASES)
def test_hip_matches_the_references_own_solver_on_multi_factor_graphs(name, dev):
    M, d = _ref_cases()
    kind, g, T0 = M.cases()[name]
    prm = _ref_prm(kind, g)
    T1, (dx1,) = _run_hip(kind, T0.astype(np.float32).copy(), g, prm, dev, max_iter=1)
    assert np.abs(dx1 - d[name + "_dx1"]).max() <= 2e-4 * np.abs(d[name + "_dx1"]).max(), np.abs(dx1 - d[name + "_dx1"]).max()
    assert np.abs(T1 - d[name + "_T1"]).max() <= 2e-5
    T10, (dx10,) = _run_hip(kind, T0.astype(np.float32).copy(), g, prm, dev, max_iter=10)
    assert np.array_equal(T10[0], T0.astype(np.float32)[0])
    assert np.abs(T10 - d[name + "_T10"]).max() <= 2e-5, np.abs(T10 - d[name + "_T10"]).max()
    assert np.abs(dx10 - d[name + "_dx10"]).max() <= 1e-5


# ------------------------------------------------------------------------------------------ CPU: the oracle itself
@pytest.mark.parametrize("kind", KINDS)
def test_oracle_jacobians_match_finite_differences(kind):
    c = S.calib_keyframe_graph(seed=2, height=12, width=16, fx=18.0)
    prm = dict(_calib_prm(c), K=c["K"].astype(np.float64))
    T = _perturb(c["T_gt"], 0, 0.05).astype(np.float64)
    e = 1
    ie, je, idx = np.array([c["ii"][e]]), np.array([c["jj"][e]]), c["idx"][e:e + 1]
    ix, jx = int(ie[0]), int(je[0])
    tij, qij, sij = G.rel_sim3(T[ix, 0:3], T[ix, 3:7], T[ix, 7], T[jx, 0:3], T[jx, 3:7], T[jx, 7])
    P = G.act_so3(qij, c["Xs"][jx].astype(np.float64)) * sij + tij
    Xi = c["Xs"][ix][idx[0]].astype(np.float64)
    rows = G._rows(kind, P, Xi, idx[0], np.ones(len(P), bool), np.ones(len(P)), prm)
    Jj = np.stack([G.apply_sim3_adj_inv(T[ix, 0:3], T[ix, 3:7], T[ix, 7], J0) for J0, _, _ in rows], 1).reshape(-1, 7)

    def left_mul(xi, pose):  # exp(xi) * pose to first order in the translation part (enough for a central difference)
        th = np.linalg.norm(xi[3:6])
        dq = np.array([0, 0, 0, 1.0]) if th < 1e-14 else np.concatenate([np.sin(th / 2) * xi[3:6] / th, [np.cos(th / 2)]])
        ds = np.exp(xi[6])
        out = np.empty(8)
        out[0:3] = G.act_so3(dq, pose[None, 0:3])[0] * ds + xi[0:3]
        out[3:7] = G.quat_comp(dq, pose[3:7])
        out[7] = ds * pose[7]
        return out

    h = 1e-6
    for k, sign in ((jx, 1.0), (ix, -1.0)):  # J_i = -J_j (gn_kernels.cu:606)
        Jn = np.zeros_like(Jj)
        for d in range(7):
            xi = np.zeros(7)
            xi[d] = h
            Tp, Tm = T.copy(), T.copy()
            Tp[k], Tm[k] = left_mul(xi, T[k]), left_mul(-xi, T[k])
            Jn[:, d] = (G.residual_vector(kind, Tp, c["Xs"], ie, je, idx, prm) - G.residual_vector(kind, Tm, c["Xs"], ie, je, idx, prm)) / (2 * h)
        m = np.isfinite(Jn).all(1) & np.isfinite(Jj).all(1)
        assert m.mean() > 0.9
        assert np.abs(sign * Jj[m] - Jn[m]).max() <= 1e-5 * max(1.0, np.abs(Jn[m]).max())


def test_oracle_exp_sim3_identities():
    t, q, s = G.exp_sim3(np.zeros(7, dtype=np.float32))
    assert np.array_equal(t, np.zeros(3)) and np.array_equal(q, [0, 0, 0, 1]) and s == 1
    rng = np.random.default_rng(0)
    for mag in (1e-4, 0.3):  # series branch and closed-form branch: exp(xi) exp(-xi) = identity
        xi = (mag * rng.standard_normal(7)).astype(np.float32)
        ident = np.array([0, 0, 0, 0, 0, 0, 1, 1], dtype=np.float32)
        back = G.retr_sim3(-xi, G.retr_sim3(xi, ident))
        assert np.abs(back - ident).max() < 5e-6


@pytest.mark.parametrize("kind", ("points", "rays"))
def test_oracle_recovers_ground_truth(kind):
    g = S.keyframe_graph(num_poses=4, n=300, seed=1)
    T = _perturb(g["T_gt"], 5, 0.03)
    G.gauss_newton(kind, T, g["Xs"], g["Cs"], g["ii"], g["jj"], g["idx"], g["valid"], g["Q"], PRM, 10, 1e-8)
    assert np.abs(T - g["T_gt"]).max() < 2e-6


# ------------------------------------------------------------------------------------------ GPU: HIP vs oracle
def _run_hip(kind, T, g, prm, dev, max_iter=10, delta=1e-8, debug=False):
    import artdeco_amd
    artdeco_amd.install_dropins()
    import mast3r_slam_backends as B
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    Twc = t(T)
    common = (t(g["Xs"]), t(g["Cs"]))
    graph = (t(g["ii"]), t(g["jj"]), t(g["idx"]), t(g["valid"]), t(g["Q"]))
    if debug:
        k = KINDS.index(kind)
        sa, sb = {"points": (prm["sigma_point"], 1.0), "rays": (prm["sigma_ray"], prm["sigma_dist"]),
                  "calib": (prm.get("sigma_pixel"), prm.get("sigma_depth"))}[kind]
        out = B._gauss_newton(k, Twc, *common, t(g["K"]) if kind == "calib" else None, *graph, g.get("height", 0), g.get("width", 0),
                              prm.get("pixel_border", 0), prm.get("z_eps", 0.0), sa, sb, prm["C_thresh"], prm["Q_thresh"], max_iter,
                              delta, debug_blocks=True)
    elif kind == "points":
        out = B.gauss_newton_points(Twc, *common, *graph, prm["sigma_point"], prm["C_thresh"], prm["Q_thresh"], max_iter, delta)
    elif kind == "rays":
        out = B.gauss_newton_rays(Twc, *common, *graph, prm["sigma_ray"], prm["sigma_dist"], prm["C_thresh"], prm["Q_thresh"], max_iter, delta)
    else:
        out = B.gauss_newton_calib(Twc, *common, t(g["K"]), *graph, g["height"], g["width"], prm["pixel_border"], prm["z_eps"],
                                   prm["sigma_pixel"], prm["sigma_depth"], prm["C_thresh"], prm["Q_thresh"], max_iter, delta)
    return Twc.cpu().numpy(), [o.cpu().numpy() for o in out]


@pytest.mark.gpu
@pytest.mark.parametrize("kind", KINDS)
def test_factor_blocks_match_oracle(kind, dev):
    """The reference's intermediate (Hs [4,E,7,7], gs [2,E,7], gn_kernels.cu:713-745) at the initial poses."""
    g, prm = _graph(kind, seed=3)
    T0 = _perturb(g["T_gt"], 1, 0.02)
    _, (dx, Hs, gs) = _run_hip(kind, T0.copy(), g, prm, dev, max_iter=1, debug=True)
    uniq = np.unique(np.concatenate([g["ii"], g["jj"]]))
    Ho, go = G.factor_blocks(kind, T0, g["Xs"], g["Cs"], np.searchsorted(uniq, g["ii"]), np.searchsorted(uniq, g["jj"]),
                             g["idx"], g["valid"], g["Q"], prm)
    for e in range(len(g["ii"])):
        for b in range(4):
            assert np.abs(Hs[b, e] - Ho[b, e]).max() <= 2e-4 * np.abs(Ho[:, e]).max(), (kind, e, b)
        for b in range(2):
            assert np.abs(gs[b, e] - go[b, e]).max() <= 2e-4 * max(np.abs(go[:, e]).max(), 1e-3 * np.abs(Ho[:, e]).max()), (kind, e, b)
    dxo = G.solve_step(Ho, go, np.searchsorted(uniq, g["ii"]), np.searchsorted(uniq, g["jj"]), len(T0))
    assert np.abs(dx - dxo).max() <= 2e-4 * max(np.abs(dxo).max(), 1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", KINDS)
def test_full_solve_matches_oracle(kind, dev):
    g, prm = _graph(kind, seed=4)
    T0 = _perturb(g["T_gt"], 2, 0.02)
    To = T0.copy()
    dxo = G.gauss_newton(kind, To, g["Xs"], g["Cs"], g["ii"], g["jj"], g["idx"], g["valid"], g["Q"], prm, 10, 1e-8)
    Th, (dxh,) = _run_hip(kind, T0.copy(), g, prm, dev)
    assert np.array_equal(Th[0], T0[0])                      # the fixed keyframe is never touched
    assert np.abs(Th - To).max() <= 2e-5, np.abs(Th - To).max()
    assert np.abs(dxh - dxo).max() <= 1e-5
    if kind != "calib":  # exact data: ground truth is the optimum (the calib graph's nearest-pixel matches bias its optimum)
        assert np.abs(Th - g["T_gt"]).max() < 5e-6


@pytest.mark.gpu
def test_keyframe_ids_need_not_be_contiguous(dev):
    a = S.keyframe_graph(num_poses=4, n=400, seed=6)
    b = S.keyframe_graph(num_poses=4, n=400, seed=6, kf_ids=[3, 10, 11, 42])
    T0 = _perturb(a["T_gt"], 3, 0.02)
    Ta, _ = _run_hip("rays", T0.copy(), a, PRM, dev)
    Tb, _ = _run_hip("rays", T0.copy(), b, PRM, dev)
    assert np.array_equal(Ta, Tb)


@pytest.mark.gpu
def test_singular_system_leaves_poses_untouched(dev):
    """No usable match (all Q below the threshold): A = 0, the factorisation fails, dx = 0 (gn_kernels.cu:155-158)."""
    g = S.keyframe_graph(num_poses=3, n=256, seed=7)
    g["Q"][:] = 1.0
    T0 = _perturb(g["T_gt"], 4, 0.02)
    Th, (dx,) = _run_hip("points", T0.copy(), g, PRM, dev)
    assert np.array_equal(Th, T0) and not dx.any()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ("points", "rays"))
def test_full_size_graph_recovers_ground_truth(kind, dev):
    """Reference size: 196 608 points per factor (512x384), 6 keyframes, 16 factors; exact data => exact recovery, and
    a second call from the solution does not move (idempotence)."""
    g = S.keyframe_graph(num_poses=6, n=512 * 384, seed=8, extra_edges=3)
    T0 = _perturb(g["T_gt"], 5, 0.03)
    Th, (dx,) = _run_hip(kind, T0.copy(), g, PRM, dev)
    assert np.abs(Th - g["T_gt"]).max() < 5e-6
    Th2, _ = _run_hip(kind, Th.copy(), g, PRM, dev)
    assert np.abs(Th2 - Th).max() < 2e-6


@pytest.mark.gpu
def test_outliers_are_downweighted_by_huber(dev):
    g = S.keyframe_graph(num_poses=4, n=4096, seed=9, outlier_frac=0.15)
    T0 = _perturb(g["T_gt"], 6, 0.02)
    Th, _ = _run_hip("rays", T0.copy(), g, PRM, dev)
    To = T0.copy()
    G.gauss_newton("rays", To, g["Xs"], g["Cs"], g["ii"], g["jj"], g["idx"], g["valid"], g["Q"], PRM, 10, 1e-8)
    assert np.abs(Th - To).max() <= 5e-5
    assert np.abs(Th - g["T_gt"]).max() < 0.02 and np.abs(Th - g["T_gt"]).max() < np.abs(T0 - g["T_gt"]).max()

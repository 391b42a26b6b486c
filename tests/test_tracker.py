"""Frontend Sim(3) tracker (CameraTracker.track).

CPU (-m "not gpu"):
  * oracle/tracker_oracle.py against tests/golden/tracker_*.npz -- outputs of the REFERENCE's own CameraTracker.track()
    executed on CPU (tests/golden/make_golden_tracker.py): every iteration's tau and cost, the final pose, the three
    flags, last_dist, the fused keyframe pointmap;
  * the kernels' arithmetic header (artdeco_amd/csrc/tracker_math.hpp) compiled for the host and driven through the same
    launch sequence (tests/host/tracker_host.cpp) against the oracle and the goldens: everything but the GPU plumbing.
GPU (-m gpu): the HIP path through the C ABI against the oracle and the goldens (same checks), the CameraTracker class end
to end, run-to-run bit reproducibility, and the reference's full frame size (512x384).
"""
import ctypes
import os
import subprocess
import types

import numpy as np
import pytest
import torch

from artdeco_amd import synthetic as S
from oracle import tracker_oracle as TO

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
CASES = ["tracker_cov", "tracker_nocov", "tracker_moved_kf", "tracker_ragged", "tracker_rough", "tracker_kfN3", "tracker_baddepth",
         "tracker_far", "tracker_newkf", "tracker_lost"]
FOCAL_CASES = ["tracker_focal", "tracker_focal_nocov"]     # --optimize_focal: goldens from the reference run with args.optimize_focal = True
CFG = dict(TO.BASE_CFG)


def load_case(name):
    d = np.load(os.path.join(GOLD, name + ".npz"))
    kw = eval(str(d["scene_kw"]))  # repr of a dict of numbers / lists written by make_golden_tracker.py
    if "kf_pose" in kw:
        kw["kf_pose"] = np.array(kw["kf_pose"])
    sc = S.tracker_scene(**kw)
    for k, v in sc.items():
        if isinstance(v, np.ndarray):  # the seeded generator still produces the inputs the golden was made from
            assert float(d["in_sum_" + k]) == float(np.asarray(v, dtype=np.float64).sum()), k
    return sc, d


def is_focal(d):
    return "optimize_focal" in d.files and bool(d["optimize_focal"])


def case_K(sc, d):
    """The K the tracker was handed (the focal cases start from a deliberately wrong focal)."""
    return np.ascontiguousarray(d["K_in"] if is_focal(d) else sc["K"], dtype=np.float32)


def oracle_run(sc, d, det_mode, trace=None):
    return TO.track(sc, covariance_filter=bool(d["covariance_filter"]), min_displacement=float(d["min_displacement"]),
                    thres_keyframe=float(d["thres_keyframe"]), last_dist=float(d["last_dist_in"]), det_mode=det_mode, trace=trace,
                    optimize_focal=is_focal(d), K=case_K(sc, d))


# ---------------------------------------------------------------------------------- CPU: oracle vs the reference's outputs
@pytest.mark.parametrize("det_mode", ["lu", "analytic"])
@pytest.mark.parametrize("name", CASES + FOCAL_CASES)
def test_oracle_matches_reference_goldens(name, det_mode):
    sc, d = load_case(name)
    tr = []
    o = oracle_run(sc, d, det_mode, tr)
    assert (o["lost"], o["is_keyframe"], o["is_keyframe_map"]) == tuple(bool(x) for x in d["out_flags"])
    assert len(tr) == len(d["out_costs"])
    if tr:
        taus = np.stack([t["tau"] for t in tr])
        costs = np.array([t["cost"] for t in tr])
        assert np.abs(taus - d["out_taus"]).max() < 5e-4 * max(1.0, np.abs(d["out_taus"]).max())
        assert np.abs(costs / d["out_costs"] - 1).max() < 2e-4
    assert np.abs(o["T_WCf"] - d["out_T_WCf"][0]).max() < 2e-5
    assert abs(float(o["last_dist"]) - float(d["out_last_dist"])) < 1e-5
    assert int(o["kf_N"]) == int(d["out_kf_N"])
    assert np.abs(o["kf_X"] - d["out_kf_X"]).max() < 5e-5
    assert np.abs(o["kf_C"] - d["out_kf_C"]).max() < 1e-5
    if name == "tracker_rough":  # the 0.9-quantile branch of the covariance filter is what this case is for
        assert all(t["thr"] > 1.0 for t in tr)


def test_oracle_quantile_is_torch_quantile():
    rng = np.random.default_rng(0)
    for n in (1, 2, 3, 10, 1000, 4097):
        for q in (0.0, 0.5, 0.8, 0.9, 1.0):
            x = rng.standard_normal(n).astype(np.float32)
            if n > 5:
                x[rng.integers(0, n, n // 3)] = x[0]  # ties
            ref = torch.quantile(torch.from_numpy(x), q).item()
            assert float(TO.quantile_linear(x, q)) == ref, (n, q)


# ---------------------------------------------------------------------------------- backends: host harness / HIP
@pytest.fixture(scope="session")
def host():
    """tests/host/tracker_host.cpp: the kernels' arithmetic header compiled with g++ (test infrastructure only)."""
    src = os.path.join(ROOT, "tests", "host", "tracker_host.cpp")
    out_dir = os.path.join(ROOT, "tests", "host", "_build")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, "libtracker_host.so")
    hdr = os.path.join(ROOT, "artdeco_amd", "csrc", "tracker_math.hpp")
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.run(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-I", os.path.join(ROOT, "artdeco_amd", "csrc"), src, "-o", out],
                       check=True)
    lib = ctypes.CDLL(out)
    lib.th_quantile.restype = ctypes.c_float
    lib.th_quantile.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_float]
    return lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def run_host(host, sc, cov, thres, cfg=CFG, max_iters=None, focal=False, K=None):
    n = sc["height"] * sc["width"]
    f = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    arrs = dict(K=f(sc["K"] if K is None else K), Xf=f(sc["Xff"]), Cf=f(sc["Cff"]), Qf=f(sc["Qff"]), Xk=f(sc["Xk_canon"]), Ck=f(sc["Ck"]), Qk=f(sc["Qkf"]),
                idx=np.ascontiguousarray(sc["idx_f2k"]), vm=np.ascontiguousarray(sc["valid_match"].astype(np.uint8)),
                Tf=f(sc["T_WCf0"]), Tk=f(sc["T_WCk"]))
    res = np.zeros(32, np.float32)
    mi = int(cfg["max_iters"] if max_iters is None else max_iters)
    dbg = dict(Xc=np.zeros((n, 3), np.float32), var=np.zeros((n, 3), np.float32), valid_opt=np.zeros(n, np.uint8), acc0=np.zeros(45, np.float32),
               thr=np.zeros(max(mi, 1), np.float32))
    c_f = ctypes.c_float
    rc = host.th_track_frame(sc["height"], sc["width"], _ptr(arrs["K"]), _ptr(arrs["Xf"]), _ptr(arrs["Cf"]), c_f(1.0), _ptr(arrs["Qf"]),
                             _ptr(arrs["Xk"]), _ptr(arrs["Ck"]), c_f(1.0 / sc["kf_N"]), _ptr(arrs["Qk"]), _ptr(arrs["idx"]), _ptr(arrs["vm"]),
                             _ptr(arrs["Tf"]), _ptr(arrs["Tk"]), c_f(cfg["sigma_pixel"]), c_f(cfg["sigma_depth"]), c_f(cfg["huber"]),
                             c_f(cfg["C_conf"]), c_f(cfg["Q_conf"]), c_f(cfg["min_match_frac"]), int(cfg["pixel_border"]),
                             c_f(cfg["depth_eps"]), c_f(cfg["rel_error"]), c_f(cfg["delta_norm"]), mi, int(cov), int(focal), c_f(thres), _ptr(res),
                             _ptr(dbg["Xc"]), _ptr(dbg["var"]), _ptr(dbg["valid_opt"]), _ptr(dbg["acc0"]), _ptr(dbg["thr"]))
    assert rc == 0
    return res, dbg


def run_hip(dev, sc, cov, thres, cfg=CFG, max_iters=None, chunk=None, focal=False, K=None):
    from artdeco_amd import tracker as T
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    cfg = dict(cfg, max_iters=cfg["max_iters"] if max_iters is None else max_iters)
    res, dbg = T.track_frame(sc["height"], sc["width"], t(sc["K"] if K is None else K), t(sc["Xff"]), t(sc["Cff"]), 1, t(sc["Qff"]), t(sc["Xk_canon"]),
                             t(sc["Ck"]), sc["kf_N"], t(sc["Qkf"]), t(sc["idx_f2k"]), t(sc["valid_match"]), t(sc["T_WCf0"]), t(sc["T_WCk"]), cfg,
                             covariance_filter=cov, thres_keyframe=thres, debug=True, chunk=chunk, optimize_focal=focal)
    torch.cuda.synchronize()
    return res.cpu().numpy(), {k: v.cpu().numpy() for k, v in dbg.items()}


def unpack_acc(acc, nv=7):
    H = np.zeros((nv, nv))
    l = 0
    for n in range(nv):
        for m in range(n + 1):
            H[n, m] = H[m, n] = acc[l]
            l += 1
    return H, -np.asarray(acc[l:l + nv], dtype=np.float64), float(acc[l + nv])  # (H, g = -J^T e, cost) in the reference's convention


def check_against_oracle_and_golden(name, res, dbg):
    sc, d = load_case(name)
    n = sc["height"] * sc["width"]
    tr = []
    o = oracle_run(sc, d, "analytic", tr)
    # stage outputs
    assert np.abs(dbg["Xc"] - o["Xf_c"]).max() < 1e-6
    # var = E[x^2] - E[x]^2 in float32 (the reference's formula): the absolute error scales with x^2, not with var
    assert np.abs(dbg["var"] - o["var_f"]).max() <= 4e-6 * max(1.0, float((o["Xf_c"] ** 2).max()))
    assert np.array_equal(dbg["valid_opt"].astype(bool), o["valid_opt"])
    assert int(res[19]) == int(o["valid_opt"].sum()) and int(res[20]) == int(o["valid_kf"].sum())
    vm = sc["valid_match"][:, 0]
    assert int(res[21]) == len(np.unique(sc["idx_f2k"][vm]))
    assert bool(res[16]) == bool(d["out_flags"][0]) and not bool(res[17])
    if bool(res[16]):  # lost: nothing else is defined, the pose is handed back untouched
        assert np.array_equal(res[0:8], sc["T_WCf0"][0])
        return
    # first linearisation: H, g, cost
    H, g, cost = unpack_acc(dbg["acc0"], 8 if is_focal(d) else 7)
    Ho, go, co = tr[0]["H"], tr[0]["g"], tr[0]["cost"]
    assert np.abs(H - Ho).max() < 2e-4 * np.abs(Ho).max()
    assert np.abs(g - go).max() < 2e-4 * np.abs(go).max()
    assert abs(cost / co - 1) < 1e-4
    # the solve
    assert int(res[18]) == o["iterations"] == len(d["out_costs"])
    assert np.abs(res[0:8] - o["T_WCf"]).max() < 3e-5
    assert np.abs(res[8:16] - o["T_CkCf"]).max() < 3e-5
    assert np.abs(res[0:8] - d["out_T_WCf"][0]).max() < 5e-5          # ... and the reference itself
    assert abs(res[23] / d["out_costs"][-1] - 1) < 3e-4
    if is_focal(d):   # the focal the iterations ended with, against the oracle's and the reference's updated K_slam
        assert res[26] == res[27]
        assert abs(res[26] - o["K"][0, 0]) < 2e-5 * o["K"][0, 0] and abs(res[26] - d["out_K"][0, 0]) < 2e-5 * d["out_K"][0, 0]
    else:
        assert res[26] == np.float32(sc["K"][0, 0]) and res[27] == np.float32(sc["K"][1, 1])
    # keyframe decisions from the device counts
    from artdeco_amd.tracker import TrackOutcome, keyframe_decisions
    oc = TrackOutcome(None, None, False, False, int(res[18]), int(res[19]), int(res[20]), int(res[21]), float(res[22]), float(res[23]))
    is_kf, is_map, last = keyframe_decisions(oc, n, CFG["match_frac_thresh"], float(d["min_displacement"]), float(d["last_dist_in"]))
    assert (is_kf, is_map) == (bool(d["out_flags"][1]), bool(d["out_flags"][2]))
    assert abs(float(last) - float(d["out_last_dist"])) < 1e-5
    if "dist_quantile" in o:
        assert float(res[22]) == np.float32(o["dist_quantile"])      # order statistics are exact


# ---------------------------------------------------------------------------------- CPU: the kernels' arithmetic on the host
@pytest.mark.parametrize("name", CASES + FOCAL_CASES)
def test_host_compiled_kernel_math(name, host):
    sc, d = load_case(name)
    res, dbg = run_host(host, sc, bool(d["covariance_filter"]), float(d["thres_keyframe"]), focal=is_focal(d), K=case_K(sc, d))
    check_against_oracle_and_golden(name, res, dbg)
    if name == "tracker_rough":
        tr = []
        oracle_run(sc, d, "analytic", tr)
        thr = dbg["thr"][:len(tr)]
        assert np.abs(thr / np.array([t["thr"] for t in tr]) - 1).max() < 1e-5 and (thr > 1).all()


def test_host_radix_select_is_torch_quantile(host):
    rng = np.random.default_rng(1)
    for n in (1, 2, 7, 1000, 70001):
        for q in (0.0, 0.3, 0.8, 0.9, 1.0):
            x = (rng.standard_normal(n) * 10 ** rng.uniform(-3, 3)).astype(np.float32)
            if n > 5:
                x[rng.integers(0, n, n // 4)] = x[1]
                x[rng.integers(0, n, 3)] = np.inf
                x[rng.integers(0, n, 3)] = -0.0
            got = host.th_quantile(_ptr(x), n, ctypes.c_float(q))
            ref = torch.quantile(torch.from_numpy(x), q).item()   # lerp(inf, inf) is nan in torch too
            assert got == ref or (np.isnan(got) and np.isnan(ref)), (n, q)


def test_host_retraction_matches_oracle(host):
    rng = np.random.default_rng(2)
    T = S.random_poses(rng, 3)[1:]
    for k, tau in enumerate([np.zeros(7), 1e-9 * np.ones(7), np.array([0.1, -0.2, 0.05, 0.02, 0.03, -0.01, 0.0]),
                             np.array([0, 0, 0, 0, 0, 0, 0.1]), 0.3 * rng.standard_normal(7)]):
        tau = tau.astype(np.float32)
        out = np.zeros(8, np.float32)
        host.th_exp_retract(_ptr(tau), _ptr(np.ascontiguousarray(T[k % 2])), _ptr(out))
        ref = TO.quat2unit(TO.sim3_retract(tau, T[k % 2]))
        assert np.abs(out - ref).max() < 2e-6


def test_host_fusion_matches_reference(host):
    sc, d = load_case("tracker_cov")
    res, _ = run_host(host, sc, True, 0.8)
    X, C = sc["Xk_canon"].copy(), sc["Ck"][:, 0].copy()
    host.th_fuse_pointmap(ctypes.c_int64(len(X)), _ptr(res), _ptr(np.ascontiguousarray(sc["Xkf"])), _ptr(np.ascontiguousarray(sc["Ckf"][:, 0])),
                          _ptr(X), _ptr(C))
    assert np.abs(X - d["out_kf_X"]).max() < 5e-5 and np.abs(C - d["out_kf_C"][:, 0]).max() < 1e-6


def test_host_full_frame_matches_oracle(host):
    """512x384 (the reference's SLAM resolution): 196 608 matches through the two-level select and the accumulators."""
    sc = S.tracker_scene(height=384, width=512, seed=12, fx=420.0, pose_noise=0.04, rough_cols=0.25)
    res, dbg = run_host(host, sc, True, 0.8)
    tr = []
    o = TO.track(sc, None, True, det_mode="analytic", trace=tr)
    assert int(res[18]) == o["iterations"] and not bool(res[16]) and not bool(res[17])
    assert np.abs(res[0:8] - o["T_WCf"]).max() < 5e-5
    assert np.array_equal(dbg["valid_opt"].astype(bool), o["valid_opt"])
    assert float(res[22]) == np.float32(o["dist_quantile"])
    thr = dbg["thr"][:len(tr)]
    assert (thr > 1).all() and thr[0] == np.float32(tr[0]["thr"]) and np.abs(thr / np.array([t["thr"] for t in tr]) - 1).max() < 1e-4


def test_library_exports_tracker_symbols(lib):
    for sym in ("adk_track_workspace_bytes", "adk_track_frame", "adk_track_fuse_pointmap"):
        assert hasattr(lib, sym)
    assert lib.adk_track_workspace_bytes(384, 512) > 0
    assert lib.adk_track_workspace_bytes(0, 512) == -1


def test_tracker_rejects_cpu_tensors():
    from artdeco_amd import _lib, tracker as T
    sc = S.tracker_scene(height=8, width=8)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    with pytest.raises(_lib.AdkError):
        T.track_frame(8, 8, t(sc["K"]), t(sc["Xff"]), t(sc["Cff"]), 1, t(sc["Qff"]), t(sc["Xk_canon"]), t(sc["Ck"]), 1, t(sc["Qkf"]),
                      t(sc["idx_f2k"]), t(sc["valid_match"]), t(sc["T_WCf0"]), t(sc["T_WCk"]), CFG)


# ---------------------------------------------------------------------------------- GPU: the HIP path
@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES + FOCAL_CASES)
def test_hip_tracker_matches_oracle_and_reference(name, dev, lib):
    sc, d = load_case(name)
    res, dbg = run_hip(dev, sc, bool(d["covariance_filter"]), float(d["thres_keyframe"]), focal=is_focal(d), K=case_K(sc, d))
    check_against_oracle_and_golden(name, res, dbg)


@pytest.mark.gpu
def test_hip_tracker_is_bit_reproducible(dev, lib):
    sc, d = load_case("tracker_rough")
    a, _ = run_hip(dev, sc, True, 0.8)
    b, _ = run_hip(dev, sc, True, 0.8)
    assert np.array_equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tracker_far", "tracker_rough", "tracker_lost"])
def test_hip_tracker_chunked_iterations_are_identical(name, dev, lib):
    """Enqueueing the iterations in chunks (host reads result[24] in between, resume = 1) is the same computation."""
    sc, d = load_case(name)
    cov = bool(d["covariance_filter"])
    full, _ = run_hip(dev, sc, cov, 0.8)
    for chunk in (1, 2, 6):
        part, _ = run_hip(dev, sc, cov, 0.8, chunk=chunk)
        assert np.array_equal(full, part), chunk
    assert full[24] == 1.0


@pytest.mark.gpu
def test_hip_tracker_iteration_cap_and_zero_iterations(dev, lib):
    sc, d = load_case("tracker_far")
    r0, _ = run_hip(dev, sc, True, 0.8, max_iters=0)
    assert int(r0[18]) == 0 and np.abs(r0[0:8] - TO.quat2unit(sc["T_WCf0"][0])).max() < 5e-6
    r1, _ = run_hip(dev, sc, True, 0.8, max_iters=1)
    assert int(r1[18]) == 1
    tr = []
    TO.track(sc, dict(max_iters=1), True, det_mode="analytic", trace=tr)
    assert abs(r1[23] / tr[0]["cost"] - 1) < 1e-4


class _Pose:  # duck-typed stand-in for a pypose Sim3 (pypose is not installed here)
    def __init__(self, data):
        self.data = data

    def tensor(self):
        return self.data

    def to(self, dev):
        return _Pose(self.data.to(dev))


class _Frame:  # the ImageFrame fields the tracker touches (VSLAM/ImageFrame.py:15-52)
    def __init__(self, frame_id, T_WC, dev):
        self.frame_id, self.T_WC, self.dev = frame_id, T_WC, dev
        self.X_canon = self.C = None
        self.N = self.N_updates = 0

    def update_pointmap(self, X, C):
        if self.N == 0:
            self.X_canon, self.C, self.N, self.N_updates = X.clone(), C.clone(), 1, 1
            return
        self.X_canon = ((self.C * self.X_canon) + (C * X)) / (self.C + C)
        self.C = self.C + C
        self.N += 1
        self.N_updates += 1

    def to(self, dev):
        return self


class _Keyframes(list):
    def last_keyframe(self):
        return self[-1]


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES + FOCAL_CASES)
def test_hip_camera_tracker_class_end_to_end(name, dev, lib):
    """CameraTracker.track() as VSLAM/Frontend.py:80 calls it, against the reference's own run of the same frame."""
    from artdeco_amd.tracker import CameraTracker
    sc, d = load_case(name)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    H, W = sc["height"], sc["width"]
    keyframe = _Frame(0, _Pose(t(sc["T_WCk"])), dev)
    keyframe.update_pointmap(t(sc["Xk_canon"]), t(sc["Ck"]))
    keyframe.N = keyframe.N_updates = sc["kf_N"]
    frame = _Frame(1, _Pose(t(sc["T_WCf0"])), dev)
    kfs = _Keyframes([keyframe])

    def match(config, model, frame_i, frame_j, idx_i2j_init=None, embeddings_j=None):
        return (t(sc["idx_f2k"])[None], t(sc["valid_match"])[None], t(sc["Xff"]), t(sc["Cff"]), t(sc["Qff"]), t(sc["Xkf"]), t(sc["Ckf"]),
                t(sc["Qkf"]), "featf", "posf")

    args = types.SimpleNamespace(optimize_focal=is_focal(d), covariance_filter=bool(d["covariance_filter"]), point_fusion_frontend=True)
    K_slam = t(case_K(sc, d))
    trk = CameraTracker(args, {"tracking": CFG}, float(d["min_displacement"]), float(d["thres_keyframe"]), None, kfs, H, W, K_slam, dev,
                        match_fn=match)
    trk.last_dist = float(d["last_dist_in"])
    trk.last_embedding = [None, None]
    flags = trk.track(frame)
    assert tuple(bool(x) for x in flags) == tuple(bool(x) for x in d["out_flags"])
    assert isinstance(frame.T_WC, _Pose)
    assert np.abs(frame.T_WC.tensor().reshape(-1).cpu().numpy() - d["out_T_WCf"][0]).max() < 5e-5
    assert abs(float(trk.last_dist) - float(d["out_last_dist"])) < 1e-5
    assert (trk.idx_f2k is None) == bool(d["out_idx_reset"])
    assert kfs[0].N == int(d["out_kf_N"])
    assert np.abs(kfs[0].X_canon.cpu().numpy() - d["out_kf_X"]).max() < 1e-4
    assert np.abs(kfs[0].C.cpu().numpy() - d["out_kf_C"]).max() < 1e-5
    if bool(d["out_flags"][1]):
        assert trk.last_embedding == ["featf", "posf"]
    if is_focal(d):    # the caller's K_slam tensor is updated in place, as the reference does (CameraTracker.py:376-377)
        assert trk.K_slam is K_slam and np.abs(K_slam.cpu().numpy() - d["out_K"]).max() < 2e-5 * float(d["out_K"][0, 0])
    else:
        assert np.array_equal(K_slam.cpu().numpy(), case_K(sc, d))


@pytest.mark.gpu
def test_hip_tracker_full_frame_recovers_pose(dev, lib):
    """The reference's frame size (512x384 = 196 608 matches): a clean scene must come back to the exact relative pose,
    and the result must agree with the oracle run on the same inputs."""
    sc = S.tracker_scene(height=384, width=512, seed=11, fx=420.0, pose_noise=0.04, depth_noise=0.0, outlier_frac=0.0)
    res, dbg = run_hip(dev, sc, True, 0.8)
    assert not bool(res[16]) and not bool(res[17]) and 2 <= int(res[18]) <= 15
    o = TO.track(sc, None, True, det_mode="analytic")
    assert int(res[18]) == o["iterations"]
    assert np.abs(res[0:8] - o["T_WCf"]).max() < 5e-5
    assert np.array_equal(dbg["valid_opt"].astype(bool), o["valid_opt"])
    assert float(res[22]) == np.float32(o["dist_quantile"])
    q = res[3:7] * np.sign(res[6]) - sc["T_WCf_gt"][0, 3:7] * np.sign(sc["T_WCf_gt"][0, 6])
    assert np.abs(q).max() < 3e-3 and np.abs(res[0:3] - sc["T_WCf_gt"][0, 0:3]).max() < 1e-2 and abs(res[7] / sc["T_WCf_gt"][0, 7] - 1) < 3e-3


REF_TRACKER = "/root/reference/VSLAM/CameraTracker.py"


@pytest.mark.skipif(not os.path.exists(REF_TRACKER), reason="reference tree not mounted")
def test_keyframe_decisions_follow_the_references_mixed_precision_comparisons():
    """check_keyframe (CameraTracker.py:159-167) compares a float32 tensor ratio, a python-float ratio and a python-float
    threshold through python's min(); keyframe_decisions must take the same branch for counts on either side of, and exactly
    at, the threshold.  The reference method is compiled from its file and run on CPU tensors."""
    import ast
    from artdeco_amd.tracker import TrackOutcome, keyframe_decisions
    tree = ast.parse(open(REF_TRACKER).read())
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "CameraTracker")
    fn = next(n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "check_keyframe")
    ns = {"torch": torch}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), REF_TRACKER, "exec"), ns)
    rng = np.random.default_rng(0)
    for n, thr in ((3000, 0.333), (3072, 0.333), (1000, 0.25), (999, 1.0 / 3.0), (196608, 0.333)):
        stub = types.SimpleNamespace(cfg={"match_frac_thresh": thr})
        edge = int(round(thr * n))
        trials = [(edge + a, edge + b) for a in (-2, -1, 0, 1, 2) for b in (-2, -1, 0, 1, 2)] + \
                 [tuple(rng.integers(1, n, 2)) for _ in range(20)]
        for n_kf, n_unique in trials:
            n_kf, n_unique = int(np.clip(n_kf, 0, n)), int(np.clip(n_unique, 1, n))
            valid_kf = torch.zeros(n, 1, dtype=torch.bool)
            valid_kf[:n_kf] = True
            valid_match = torch.zeros(n, 1, dtype=torch.bool)
            valid_match[:n_unique] = True
            idx = torch.arange(n)                        # n_unique distinct frame pixels among the valid matches
            ref = bool(ns["check_keyframe"](stub, idx, valid_kf, valid_match))
            oc = TrackOutcome(None, None, False, False, 3, n_kf, n_kf, n_unique, 0.0, 0.0)
            mine, _, _ = keyframe_decisions(oc, n, thr, 1e9, 0.0)
            assert mine == ref, (n, thr, n_kf, n_unique)


# ---------------------------------------------------------------------------------- CPU: edge cases of the shared arithmetic
def _oracle_vs_host(host, sc, cfg=None, cov=True, thres=0.8):
    cfg = dict(CFG, **(cfg or {}))
    res, dbg = run_host(host, sc, cov, thres, cfg=cfg)
    tr = []
    o = TO.track(sc, cfg, cov, thres_keyframe=thres, det_mode="analytic", trace=tr)
    assert bool(res[16]) == o["lost"]
    assert int(res[19]) == int(o["valid_opt"].sum()) and int(res[20]) == int(o["valid_kf"].sum())
    assert np.array_equal(dbg["valid_opt"].astype(bool), o["valid_opt"])
    if not o["lost"]:
        assert int(res[18]) == o["iterations"]
        assert np.abs(res[0:8] - o["T_WCf"]).max() < 5e-5
    return res, o


def test_host_confidence_threshold_uses_the_average_over_updates(host):
    """ImageFrame.get_average_conf (ImageFrame.py:51-52): C / N, compared with C_conf (CameraTracker.py:83-84)."""
    sc = S.tracker_scene(height=24, width=32, seed=16, kf_N=3)   # seed chosen so that no stopping test sits within 30 % of its threshold
    for c_conf in (0.0, 1.4, 1.6, 5.0):   # the summed keyframe confidence is ~3 x (1..2): its average straddles 1.4 / 1.6
        res, o = _oracle_vs_host(host, sc, dict(C_conf=c_conf))
    assert o["lost"] and bool(res[16])       # C_conf = 5: nothing is confident enough -> insufficient match


def test_host_lost_threshold_boundary(host):
    """valid_opt.sum() / numel < min_match_frac in float32 (CameraTracker.py:90-91), on either side of the boundary."""
    sc = S.tracker_scene(height=24, width=32, seed=14, drop_frac=0.5)
    n = 24 * 32
    n_opt = int(TO.track(sc, None, True, det_mode="analytic")["valid_opt"].sum())
    for frac in (np.float32(n_opt) / np.float32(n), np.nextafter(np.float32(n_opt) / np.float32(n), np.float32(1)),
                 np.float32(n_opt - 1) / np.float32(n)):
        _oracle_vs_host(host, sc, dict(min_match_frac=float(frac)))


def test_host_displacement_quantile_with_heavy_ties(host):
    """Every match displaced by the same integer offset: all order statistics tie (torch.quantile returns that value)."""
    sc = S.tracker_scene(height=24, width=32, seed=15, outlier_frac=0.0)
    H, W = 24, 32
    k = np.arange(H * W)
    u, v = np.clip(k % W + 3, 0, W - 1), np.clip(k // W + 4, 0, H - 1)
    sc["idx_f2k"] = (v * W + u).astype(np.int64)
    res, o = _oracle_vs_host(host, sc)
    assert float(res[22]) == np.float32(o["dist_quantile"])
    inner = (k % W + 3 < W) & (k // W + 4 < H)
    if o["valid_opt"][inner].mean() > 0.85:
        assert float(res[22]) == 5.0


def test_host_all_matches_invalid_is_lost_and_returns_the_input_pose(host):
    sc = S.tracker_scene(height=24, width=32, seed=16)
    sc["valid_match"][:] = False
    res, o = _oracle_vs_host(host, sc)
    assert bool(res[16]) and int(res[19]) == 0 and int(res[21]) == 0 and np.array_equal(res[0:8], sc["T_WCf0"][0])

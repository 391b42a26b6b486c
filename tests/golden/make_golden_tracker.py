"""Generate tests/golden/tracker_*.npz by running the REFERENCE's own tracker on CPU.

`VSLAM/CameraTracker.py` (`CameraTracker.track`, :53-155: get_points_poses, the validity masks, opt_pose_calib_sim3
with the covariance filter, point fusion, check_keyframe, check_keyframe_map), `VSLAM/mast3r_slam/geometry.py`,
`nonlinear_optimizer.py`, `utils_uncertainty.py` and `ImageFrame.py` are imported from /root/reference and executed
unmodified.  Three things are substituted, none of them part of the path under test:
  * `pypose` (pip dependency, not vendored, not installed) -> tests/golden/pypose_stub.py (Sim(3) algebra only);
  * `cv2`, `VSLAM.utils_mast3r`, `VSLAM.mast3r_slam.visualization_utils` (viewer / network imports) -> empty modules;
  * `mast3r_match_asymmetric` -> returns the seeded synthetic match of `artdeco_amd.synthetic.tracker_scene`
    (the network has no weights here).
`CameraTracker.solve` is wrapped to record every iteration's (tau, cost).  Build container only (needs /root/reference).

    python tests/golden/make_golden_tracker.py
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, REF)

import pypose_stub  # noqa: E402

sys.modules["pypose"] = pypose_stub
sys.modules["cv2"] = types.ModuleType("cv2")
um = types.ModuleType("VSLAM.utils_mast3r")
um.mast3r_match_asymmetric = um.mast3r_inference_mono = um.inverse_normalize = None
sys.modules["VSLAM.utils_mast3r"] = um
vu = types.ModuleType("VSLAM.mast3r_slam.visualization_utils")
vu.save_pointcloud_ply = vu.visualize_matches_corr = None
sys.modules["VSLAM.mast3r_slam.visualization_utils"] = vu

import VSLAM.CameraTracker as CT  # noqa: E402  (the reference module)
from VSLAM.ImageFrame import ImageFrame  # noqa: E402

from artdeco_amd import synthetic as S  # noqa: E402

CASES = {
    # name: (scene kwargs, covariance_filter, min_displacement, last_dist)
    "tracker_cov": (dict(height=48, width=64, seed=0, fx=70.0), True, 1.0, 0.0),
    "tracker_nocov": (dict(height=48, width=64, seed=1, fx=70.0), False, 30.0, 0.0),
    "tracker_moved_kf": (dict(height=40, width=56, seed=2, fx=60.0, pose_noise=0.05,
                              kf_pose=np.array([0.3, -0.2, 0.1, 0.05, -0.08, 0.03, 0.9949, 1.2])), True, 2.0, 0.5),
    "tracker_ragged": (dict(height=37, width=53, seed=3, fx=55.0, depth_noise=0.03, outlier_frac=0.1), True, 1.0, 0.0),
    "tracker_rough": (dict(height=48, width=64, seed=7, fx=70.0, rough_cols=0.3), True, 30.0, 0.0),
    "tracker_kfN3": (dict(height=48, width=64, seed=8, fx=70.0, kf_N=3), True, 30.0, 0.0),
    "tracker_baddepth": (dict(height=48, width=64, seed=9, fx=70.0, bad_depth_frac=0.03), True, 30.0, 0.0),
    "tracker_far": (dict(height=48, width=64, seed=4, fx=70.0, pose_noise=0.12), True, 30.0, 0.0),
    "tracker_newkf": (dict(height=48, width=64, seed=5, fx=70.0, drop_frac=0.75), True, 30.0, 0.0),
    "tracker_lost": (dict(height=48, width=64, seed=6, fx=70.0, drop_frac=0.97), True, 30.0, 0.0),
    # --optimize_focal (CameraTracker.py:308-320,367-377): the tracker is handed a K whose focal is off by focal_scale
    "tracker_focal": (dict(height=48, width=64, seed=10, fx=70.0), True, 30.0, 0.0, dict(focal_scale=0.96)),
    "tracker_focal_nocov": (dict(height=40, width=56, seed=11, fx=60.0, pose_noise=0.02), False, 30.0, 0.0, dict(focal_scale=1.03)),
}


class Keyframes(list):
    def last_keyframe(self):
        return self[-1]


def run_case(name, scene_kw, cov_filter, min_disp, last_dist, focal=None):
    if "kf_pose" in scene_kw:
        kp = scene_kw["kf_pose"].astype(np.float64)
        kp[3:7] /= np.linalg.norm(kp[3:7])
        scene_kw = dict(scene_kw, kf_pose=kp)
    sc = S.tracker_scene(**scene_kw)
    H, W = sc["height"], sc["width"]
    from VSLAM.utils_config import load_config  # the reference's own loader (float resolver for "1e-6")
    cfg = load_config(os.path.join(REF, "config", "base.yaml"))
    args = types.SimpleNamespace(optimize_focal=focal is not None, covariance_filter=cov_filter, point_fusion_frontend=True)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    K_in = sc["K"].astype(np.float32).copy()
    if focal is not None:
        K_in[0, 0] *= np.float32(focal["focal_scale"])
        K_in[1, 1] *= np.float32(focal["focal_scale"])
    K_t = t(K_in.copy())
    keyframe = ImageFrame(0, 0, 0.0, torch.zeros(3, H, W), pypose_stub.Sim3(t(sc["T_WCk"])))
    keyframe.update_pointmap(t(sc["Xk_canon"]), t(sc["Ck"]))
    keyframe.N = keyframe.N_updates = sc["kf_N"]  # Xk_canon / Ck already are the fused state of kf_N predictions
    frame = ImageFrame(1, 0, 0.1, torch.zeros(3, H, W), pypose_stub.Sim3(t(sc["T_WCf0"])))
    kfs = Keyframes([keyframe])
    trk = CT.CameraTracker(args, cfg, min_disp, 0.8, None, kfs, H, W, K_t, "cpu")
    trk.last_dist = last_dist
    trk.last_embedding = [None, None]

    def fake_match(config, model, frame_i, frame_j, idx_i2j_init=None, embeddings_j=None):
        return (t(sc["idx_f2k"])[None], t(sc["valid_match"])[None], t(sc["Xff"]), t(sc["Cff"]), t(sc["Qff"]),
                t(sc["Xkf"]), t(sc["Ckf"]), t(sc["Qkf"]), "featf", "posf")

    CT.mast3r_match_asymmetric = fake_match
    taus, costs = [], []
    inner = trk.solve

    def solve(sqrt_info, r, J):
        tau, cost = inner(sqrt_info, r, J)
        taus.append(tau.numpy().copy().reshape(-1))
        costs.append(cost)
        return tau, cost

    trk.solve = solve
    lost, is_kf, is_kf_map = trk.track(frame)
    # inputs are NOT stored: tests rebuild them with S.tracker_scene(**scene_kw) (seeded numpy); a float64 checksum per
    # array detects generator drift
    out = {"in_sum_" + k: np.float64(np.asarray(v, dtype=np.float64).sum()) for k, v in sc.items() if isinstance(v, np.ndarray)}
    out.update(scene_kw=np.array(repr({k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in scene_kw.items()})),
               height=H, width=W, covariance_filter=cov_filter, min_displacement=min_disp, last_dist_in=last_dist,
               thres_keyframe=0.8, optimize_focal=focal is not None, K_in=K_in, out_K=trk.K_slam.numpy().copy(),
               out_flags=np.array([bool(lost), bool(is_kf), bool(is_kf_map)]), out_T_WCf=frame.T_WC.tensor().numpy(),
               out_taus=np.stack(taus) if taus else np.zeros((0, 7), np.float32), out_costs=np.array(costs, dtype=np.float64), out_last_dist=np.float64(trk.last_dist),
               out_kf_X=kfs[0].X_canon.numpy(), out_kf_C=kfs[0].C.numpy(), out_kf_N=kfs[0].N,
               out_idx_reset=trk.idx_f2k is None)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "flags", lost, is_kf, is_kf_map, "iters", len(costs), "cost", costs[:1], "->", costs[-1:], "last_dist", trk.last_dist)
    err = np.abs(frame.T_WC.tensor().numpy() - sc["T_WCf_gt"]).max()
    print("   |T - T_gt|max", err, " initial", np.abs(sc["T_WCf0"] - sc["T_WCf_gt"]).max())


if __name__ == "__main__":
    only = sys.argv[1:]
    for name, case in CASES.items():
        if not only or name in only:
            run_case(name, *case)

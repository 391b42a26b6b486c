"""Generate tests/golden/gn_factor_{rays,calib}.npz: ONE factor of the global optimiser solved by the REFERENCE's own
Python code.

The backend kernels (`gn_kernels.cu`, CUDA-only, not buildable here) minimise the same residuals as the frontend's
`CameraTracker.opt_pose_ray_dist_sim3` (VSLAM/CameraTracker.py:242-290: ray + distance, `point_to_ray_dist`) and
`opt_pose_calib_sim3` (:296-396: pixel + log-depth, `project_calib`), with the same sqrt(Q)/sigma weights, Huber kernel
and left-multiplicative Sim(3) update.  For a graph with one fixed keyframe i (identity-free: any pose) and one free
keyframe j, a Gauss-Newton step of `gauss_newton_rays` / `gauss_newton_calib` on the single factor (i, j) IS a step of
those functions.  This script runs them from /root/reference (pypose -> tests/golden/pypose_stub.py, as in
make_golden_tracker.py) on the rows of a synthetic factor and stores every iteration's (tau, cost) and the final pose;
tests/test_gn.py checks oracle/gn_oracle.py against them.  Build container only.

    python tests/golden/make_golden_gn_factor.py
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

import VSLAM.CameraTracker as CT  # noqa: E402
from VSLAM.utils_config import load_config  # noqa: E402

from artdeco_amd import synthetic as S  # noqa: E402

SCENES = {"rays": dict(num_poses=2, n=1500, seed=21, extra_edges=0, noise=0.003),
          "calib": dict(num_poses=2, height=40, width=56, seed=22, extra_edges=0, fx=60.0)}
PERTURB = dict(seed=5, mag=0.05)


def factor_rows(kind):
    """The (i = 0, j = 1) factor of a two-keyframe graph, row r = point r of keyframe j."""
    g = S.calib_keyframe_graph(**SCENES[kind]) if kind == "calib" else S.keyframe_graph(**SCENES[kind])
    e = int(np.nonzero((g["ii"] == 0) & (g["jj"] == 1))[0][0])
    T0 = S.perturb_poses(g["T_gt"], np.random.default_rng(PERTURB["seed"]), PERTURB["mag"])
    return g, e, T0


def main():
    cfg = load_config(os.path.join(REF, "config", "base.yaml"))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    for kind in ("rays", "calib"):
        g, e, T0 = factor_rows(kind)
        idx, vm, Q = g["idx"][e], g["valid"][e][:, 0], g["Q"][e]
        H, W = (g["height"], g["width"]) if kind == "calib" else (1, len(idx))
        args = types.SimpleNamespace(optimize_focal=False, covariance_filter=False, point_fusion_frontend=False)
        K = t(g["K"]) if kind == "calib" else torch.eye(3)
        trk = CT.CameraTracker(args, cfg, 30.0, 0.8, None, [], H, W, K, "cpu")
        taus, costs = [], []
        inner = trk.solve

        def solve(sqrt_info, r, J):
            tau, cost = inner(sqrt_info, r, J)
            taus.append(tau.numpy().copy().reshape(-1))
            costs.append(cost)
            return tau, cost

        trk.solve = solve
        Xi = g["Xs"][0][np.where(vm, idx, 0)]          # what the factor compares against, row by row
        Xj = g["Xs"][1]
        valid = (vm & (Q[:, 0] > cfg["tracking"]["Q_conf"]))[:, None]
        T_i, T_j = pypose_stub.Sim3(t(T0[0:1])), pypose_stub.Sim3(t(T0[1:2]))
        if kind == "rays":
            T_new, _ = trk.opt_pose_ray_dist_sim3(t(Xj), t(Xi), T_j, T_i, t(Q), t(valid))
        else:
            ind = np.where(vm, idx, 0)
            uv = np.stack([ind % W, ind // W], -1).astype(np.float32)
            with np.errstate(invalid="ignore", divide="ignore"):
                meas = np.concatenate([uv, np.log(Xi[:, 2:3])], -1).astype(np.float32)
            valid_meas = Xi[:, 2:3] > cfg["tracking"]["depth_eps"]
            meas[~np.repeat(valid_meas, 3, 1)] = 0.0
            T_new, _ = trk.opt_pose_calib_sim3(t(Xj), None, None, None, T_j, T_i, t(Q), t(valid), t(meas), t(valid_meas), t(ind), (H, W))
        np.savez_compressed(os.path.join(HERE, f"gn_factor_{kind}.npz"), out_taus=np.stack(taus), out_costs=np.array(costs),
                            out_T_j=pypose_stub.quat2unit(T_new).tensor().numpy(), T0=T0, edge=e,
                            in_sum_Xs=np.float64(g["Xs"].astype(np.float64).sum()))
        print(kind, "iterations", len(costs), "costs", costs, "|tau|", [float(np.linalg.norm(x)) for x in taus])


if __name__ == "__main__":
    main()

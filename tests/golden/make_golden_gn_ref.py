"""Generate tests/golden/ref_gn.npz with the REFERENCE's own Gauss-Newton solvers: /root/reference/VSLAM/backend/src/gn_kernels.cu
(point_align_kernel / ray_align_kernel / calib_proj_kernel :455-1637, pose_retr_kernel :415, SparseBlock assembly and solve :56-157,
the iteration loops of gauss_newton_points / rays / calib_cuda) compiled for the host by oracle/ref_shim/build_ref.py
(oracle/_ref/ref_gn.so: CUDA execution model and Eigen replaced by host stand-ins, nothing else changed).  Multi-factor graphs, all
three factor kinds: what pins the `points` kind and the multi-factor assembly of oracle/gn_oracle.py and csrc/gn.hip (the
single-factor goldens gn_factor_*.npz come from the reference's Python tracker and cover rays / calib only).
Inputs are seeded (`cases()`, shared with tests/test_gn.py) and not stored.  Build container only.

    python tests/golden/make_golden_gn_ref.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from artdeco_amd import synthetic as S  # noqa: E402

PRM = dict(sigma_point=0.05, sigma_ray=0.003, sigma_dist=10.0, C_thresh=0.0, Q_thresh=1.5)   # config/base.yaml:36-51
CALIB = dict(pixel_border=-10, z_eps=1e-6, sigma_pixel=1.0, sigma_depth=10.0)


def cases():
    """name -> (kind, graph, T0): 4-5 keyframes, 7-9 factors (consecutive pairs both ways + loop edges), outliers in one of them."""
    out = {}
    g = S.keyframe_graph(num_poses=4, n=700, seed=31, extra_edges=1)
    out["points_a"] = ("points", g, S.perturb_poses(g["T_gt"], np.random.default_rng(1), 0.02))
    g = S.keyframe_graph(num_poses=5, n=500, seed=32, extra_edges=2, outlier_frac=0.1)
    out["points_b"] = ("points", g, S.perturb_poses(g["T_gt"], np.random.default_rng(2), 0.03))
    g = S.keyframe_graph(num_poses=4, n=700, seed=33, extra_edges=1, noise=0.002)
    out["rays_a"] = ("rays", g, S.perturb_poses(g["T_gt"], np.random.default_rng(3), 0.02))
    g = S.keyframe_graph(num_poses=5, n=500, seed=34, extra_edges=2, kf_ids=[2, 5, 6, 9, 30])
    out["rays_b"] = ("rays", g, S.perturb_poses(g["T_gt"], np.random.default_rng(4), 0.03))
    g = S.calib_keyframe_graph(num_poses=4, height=30, width=40, seed=35, extra_edges=1, fx=44.0)
    out["calib_a"] = ("calib", g, S.perturb_poses(g["T_gt"], np.random.default_rng(5), 0.02))
    return out


def run_reference(mod, kind, g, T, max_iter, delta=1e-8):
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    Twc = t(T.astype(np.float32).copy())
    common = (Twc, t(g["Xs"]), t(g["Cs"]))
    graph = (t(g["ii"]), t(g["jj"]), t(g["idx"]), t(g["valid"]), t(g["Q"]))
    if kind == "points":
        out = mod.gauss_newton_points(*common, *graph, PRM["sigma_point"], PRM["C_thresh"], PRM["Q_thresh"], max_iter, delta)
    elif kind == "rays":
        out = mod.gauss_newton_rays(*common, *graph, PRM["sigma_ray"], PRM["sigma_dist"], PRM["C_thresh"], PRM["Q_thresh"], max_iter, delta)
    else:
        out = mod.gauss_newton_calib(*common, t(g["K"]), *graph, int(g["height"]), int(g["width"]), CALIB["pixel_border"], CALIB["z_eps"],
                                     CALIB["sigma_pixel"], CALIB["sigma_depth"], PRM["C_thresh"], PRM["Q_thresh"], max_iter, delta)
    return Twc.numpy(), out[0].numpy()


def main():
    sys.path.insert(0, os.path.join(ROOT, "oracle", "_ref"))
    from oracle.ref_shim import build_ref
    assert build_ref.build(), "oracle/_ref is not built"
    import ref_gn
    out = {}
    for name, (kind, g, T0) in cases().items():
        T1, dx1 = run_reference(ref_gn, kind, g, T0, 1)       # one Gauss-Newton iteration: step and retracted poses
        T10, dx10 = run_reference(ref_gn, kind, g, T0, 10)    # the call global_opt.py makes (max_iter 10)
        out.update({name + "_in_sum": np.float64(g["Xs"].astype(np.float64).sum() + T0.astype(np.float64).sum()),
                    name + "_T1": T1, name + "_dx1": dx1, name + "_T10": T10, name + "_dx10": dx10})
        print(name, kind, "factors", len(g["ii"]), "|dx1|", float(np.abs(dx1).max()), "|dx10|", float(np.abs(dx10).max()),
              "err to gt after 10:", float(np.abs(T10 - g["T_gt"]).max()))
    np.savez_compressed(os.path.join(HERE, "ref_gn.npz"), **out)


if __name__ == "__main__":
    main()

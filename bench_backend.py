#!/usr/bin/env python
"""Backend: one global-optimisation call = gauss_newton_rays over a keyframe graph at the reference's
factor size (512x384 = 196 608 points per factor, config/base.yaml local_opt: 10 iterations, delta_norm 1e-8),
on one MI355X through the drop-in `mast3r_slam_backends`, next to the numpy oracle on the host.  One JSON line.

    python bench_backend.py [--keyframes 16] [--extra-edges 12] [--iters 5] [--kind rays|points] [--cpu-baseline]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import artdeco_amd  # noqa: E402

artdeco_amd.install_dropins()

HBM_PEAK_GBS = 8000.0
BYTES_PER_POINT = 45  # valid 1 + idx 8 + Xi 12 (gather) + Xj 12 + Q 4 + Ci 4 (gather) + Cj 4, per (factor, point), per iteration


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--keyframes", type=int, default=16)
    ap.add_argument("--extra-edges", type=int, default=12)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--kind", default="rays", choices=["rays", "points"])
    ap.add_argument("--cpu-baseline", action="store_true")
    ap.add_argument("--random-matches", action="store_true", help="random slot order (worst case for the gathers) instead of spatially coherent matches")
    args = ap.parse_args()
    import mast3r_slam_backends as B
    from artdeco_amd import synthetic as S
    dev = torch.device("cuda:0")
    g = S.keyframe_graph(num_poses=args.keyframes, n=512 * 384, seed=0, extra_edges=args.extra_edges, coherent=not args.random_matches)
    T0 = S.perturb_poses(g["T_gt"], np.random.default_rng(1), 0.03)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    Xs, Cs, ii, jj, idx, valid, Q = t(g["Xs"]), t(g["Cs"]), t(g["ii"]), t(g["jj"]), t(g["idx"]), t(g["valid"]), t(g["Q"])
    prm = dict(sigma_point=0.05, sigma_ray=0.003, sigma_dist=10.0, C=0.0, Q=1.5, max_iter=10, delta=1e-8)

    def call(Twc):
        if args.kind == "rays":
            return B.gauss_newton_rays(Twc, Xs, Cs, ii, jj, idx, valid, Q, prm["sigma_ray"], prm["sigma_dist"], prm["C"], prm["Q"],
                                       prm["max_iter"], prm["delta"])
        return B.gauss_newton_points(Twc, Xs, Cs, ii, jj, idx, valid, Q, prm["sigma_point"], prm["C"], prm["Q"], prm["max_iter"], prm["delta"])

    Tw = t(T0)
    call(Tw)  # warm-up (also the correctness check below)
    torch.cuda.synchronize()
    err = float(np.abs(Tw.cpu().numpy() - g["T_gt"]).max())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ms = []
    for _ in range(args.iters):
        Tw = t(T0)
        e0.record()
        call(Tw)
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    ms_call = float(np.median(ms))
    E, n = len(g["ii"]), 512 * 384
    alg_bytes = BYTES_PER_POINT * E * n * prm["max_iter"]
    out = {"metric": f"global optimisation calls/s (gauss_newton_{args.kind}, {args.keyframes} keyframes, {E} factors x {n} points, 10 GN iterations)",
           "value": 1e3 / ms_call, "unit": "calls/s", "ms_per_call": ms_call, "n_gpus": 1, "higher_is_better": True, "dtype": "f32 (normal equations f64)",
           "data": "synthetic", "pose_error_after": err,
           "config": {"workload": "exactly consistent synthetic keyframe graph (artdeco_amd.synthetic.keyframe_graph), " + ("random" if args.random_matches else "spatially coherent") + " matches", "keyframes": args.keyframes,
                      "factors": E, "points_per_factor": n, "unknowns": 7 * (args.keyframes - 1)},
           "roofline": {"bound": "hbm", "kernel": "gn_accumulate_kernel (x10) + gn_solve_kernel (x10), whole call", "achieved": alg_bytes / (ms_call * 1e-3) / 1e9,
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg_bytes / (ms_call * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                        "algorithmic_bytes": alg_bytes}}
    if args.cpu_baseline:
        from oracle import gn_oracle as G  # the CPU leg only
        sub = S.keyframe_graph(num_poses=args.keyframes, n=512 * 384 // 16, seed=0, extra_edges=args.extra_edges)
        Tc = sub["T_gt"].copy()
        t0 = time.time()
        G.gauss_newton(args.kind, Tc, sub["Xs"], sub["Cs"], sub["ii"], sub["jj"], sub["idx"], sub["valid"], sub["Q"],
                       dict(sigma_point=0.05, sigma_ray=0.003, sigma_dist=10.0, C_thresh=0.0, Q_thresh=1.5), 2, 0.0)
        dt = (time.time() - t0) * 16.0 * 5.0  # 1/16 of the points, 2 of 10 iterations
        out["cpu_baseline"] = {"value": 1.0 / dt, "unit": "calls/s", "cores": 1, "kind": "port",
                               "sample": f"numpy oracle, same graph with 1/16 of the points, 2 of 10 iterations = {dt / 80:.1f} s, scaled x80"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()

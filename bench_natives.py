#!/usr/bin/env python
"""SURVEY.md 8(d)'s stand-alone rows for the natives that are not part of the optimisation step: `simple_knn._C.distIndex2` (a5,
simple_knn.cu:468-522) at P = 10^5 / 10^6, uniform and clustered, K = 3, and `mast3r_slam_backends.iter_proj` / `refine_matches` (a9 / a10,
matching_kernels.cu:119-316 / :25-116) at the frontend's real size 1 x 384 x 512 -- each timed with HIP events on one MI355X through the
drop-in modules (the C ABI), priced against the HBM roofline by its ALGORITHMIC bytes, and with the REFERENCE's own kernel compiled for the
host (oracle/_ref: simple_knn.cu / matching_kernels.cu, g++ -O2, one thread) timed beside it on the box's host cores.  One JSON line per row.

    python bench_natives.py [--reps 20] [--no-cpu]         (about a minute; the CPU legs are bounded)

Algorithmic bytes:  distIndex2  12 P (read) + 8 P (Morton sort in / out minimum) + 8 K P (write) = 44 P at K = 3 -- the search
itself is data-dependent, so points visited / boxes scanned per query are reported too (adk_knn_index2_stats);
iter_proj  36 hw (the ray image, once) + 12 hw (pts) + 8 hw (p_init) + 8 hw + 1 hw (outputs) = 65 B per pixel;
refine_matches  48 hw (D11, fp16 x 24) + 48 hw (D21) + 16 hw (p1) + 16 hw (output) = 128 B per pixel.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import artdeco_amd  # noqa: E402

artdeco_amd.install_dropins()
HBM_PEAK_GBS = 8000.0


def _events_ms(fn, reps, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        e1.synchronize()
        ms.append(e0.elapsed_time(e1))
    ms.sort()
    return {"median_ms": ms[len(ms) // 2], "min_ms": ms[0], "reps": reps}


def _row(name, t, alg_bytes, **extra):
    gbs = alg_bytes / (t["median_ms"] * 1e-3) / 1e9
    return {"row": name, **t, "algorithmic_MB": alg_bytes / 1e6, "GB_per_s": gbs, "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS}, **extra}


def knn_rows(dev, reps, cpu, full_cpu=False):
    from simple_knn import _C as knn
    from artdeco_amd import _lib
    from test_knn import _clouds
    lib = _lib.load()
    rn = None
    if cpu:
        from oracle import ref_native as rn_   # the CHECKER / baseline only: the reference's simple_knn.cu compiled for the host
        rn = rn_ if rn_.available() else None
    out = []
    for P in (100_000, 1_000_000):
        for kind in ("uniform", "surface"):
            pts_h = _clouds(kind, P, 11)
            pts = torch.from_numpy(pts_h).to(dev)
            t = _events_ms(lambda: knn.distIndex2(pts, 3), reps)
            # the search's work, counted by the measurement variant of the same kernel (same results)
            ws = torch.empty(int(lib.adk_knn_workspace_bytes(P)), dtype=torch.uint8, device=dev)
            d = torch.zeros(P * 3, device=dev); i = torch.full((P * 3,), -1, dtype=torch.int32, device=dev)
            stats = torch.zeros(4, dtype=torch.int64, device=dev)
            _lib.check(lib.adk_knn_index2_stats(pts.data_ptr(), P, d.data_ptr(), i.data_ptr(), ws.data_ptr(), ws.numel(), stats.data_ptr(),
                                                _lib.stream_of(pts)), "adk_knn_index2_stats")
            dd, ii = knn.distIndex2(pts, 3)
            assert torch.equal(dd, d) and torch.equal(ii, i), "the counted variant must give adk_knn_index2's results"
            s = stats.cpu().tolist()
            extra = {"P": P, "K": 3, "cloud": "uniform U[0,1]^3" if kind == "uniform" else "clustered (noisy samples of 8 spheres' surfaces)",
                     "boxes_scanned_per_query": s[0] / P, "points_visited_per_query": s[1] / P, "superbox_tests_per_query": s[2] / P,
                     "box_tests_per_query": s[3] / P, "queries_per_s": P / (t["median_ms"] * 1e-3)}
            if rn is not None and (P <= 100_000 or full_cpu):     # the reference's serial box walk takes 346 s at P = 10^6 (2 893 queries/s, measured once: profiles/r06_bench_natives.jsonl)
                t0 = time.perf_counter()
                dr, ir = rn.knn_index2(pts_h, 3)
                cpu_s = time.perf_counter() - t0
                rows = np.random.default_rng(3).choice(P, 5000, replace=False)
                assert np.array_equal(np.sort(dr[rows], 1), np.sort(d.cpu().numpy().reshape(P, 3)[rows], 1)), "distances differ from the reference's"
                extra["cpu_baseline"] = {"value": P / cpu_s, "unit": "queries/s", "cores": 1, "kind": "reference",
                                         "sample": f"the reference's simple_knn.cu (SimpleKNN::knn_index2) compiled for the host, whole {P}-point call: {cpu_s:.2f} s; "
                                                   "squared distances of 5 000 rows equal to the HIP path's"}
            out.append(_row(f"simple_knn._C.distIndex2 P={P} {kind} K=3", t, 44 * P, **extra))
    return out


def matching_rows(dev, reps, cpu):
    import mast3r_slam_backends as B
    from test_matching import _full_size_inputs
    inp = _full_size_inputs()
    hw = 384 * 512
    t = lambda a: torch.from_numpy(a).to(dev)
    rays, pts, p0 = t(inp["rays"]), t(inp["pts"]), t(inp["p_init"])
    D11, D21, p1 = t(inp["D11"]), t(inp["D21"]), t(inp["p1"])
    out = []
    ti = _events_ms(lambda: B.iter_proj(rays, pts, p0, 10, 1e-8, 1e-6), reps)
    tr = _events_ms(lambda: B.refine_matches(D11, D21, p1, 4, 5), reps)
    ei, er = {"size": "1 x 384 x 512", "max_iter": 10}, {"size": "1 x 384 x 512", "radius": 4, "dilation_max": 5, "descriptors": str(D11.dtype)}
    if cpu:
        from oracle import ref_native as rn
        if rn.available():
            rm = rn.ref_matching()
            c = lambda a: torch.from_numpy(a)
            t0 = time.perf_counter(); pr, cr = rm.iter_proj(c(inp["rays"]), c(inp["pts"]), c(inp["p_init"]), 10, 1e-8, 1e-6); s_i = time.perf_counter() - t0
            t0 = time.perf_counter(); (qr,) = rm.refine_matches(c(inp["D11"]), c(inp["D21"]), c(inp["p1"]), 4, 5); s_r = time.perf_counter() - t0
            ph, ch = B.iter_proj(rays, pts, p0, 10, 1e-8, 1e-6)
            (qh,) = B.refine_matches(D11, D21, p1, 4, 5)
            assert torch.equal(ph.cpu(), pr) and torch.equal(ch.cpu(), cr) and torch.equal(qh.cpu(), qr), "HIP result differs from the reference's"
            ei["cpu_baseline"] = {"value": hw / s_i, "unit": "pixels/s", "cores": 1, "kind": "reference",
                                  "sample": f"matching_kernels.cu iter_proj_kernel compiled for the host, the whole 196 608-pixel call: {s_i:.2f} s; result bit-identical to the HIP path's"}
            er["cpu_baseline"] = {"value": hw / s_r, "unit": "pixels/s", "cores": 1, "kind": "reference",
                                  "sample": f"matching_kernels.cu refine_matches_kernel compiled for the host, the whole call: {s_r:.2f} s; result bit-identical to the HIP path's"}
    ei["pixels_per_s"], er["pixels_per_s"] = hw / (ti["median_ms"] * 1e-3), hw / (tr["median_ms"] * 1e-3)
    # what the kernels are really bound by: iter_proj re-gathers the 36 B ray texel bilinearly (4 texels) twice per iteration -> 10 x 2 x 4 x 36 B
    # of cache traffic per pixel; refine_matches reads 5 x 81 candidate descriptors of 48 B per pixel from L2
    ei["gathered_bytes_per_pixel"] = 10 * 2 * 4 * 36
    er["gathered_bytes_per_pixel"] = 5 * 81 * 48
    ei["gather_GB_per_s"] = ei["gathered_bytes_per_pixel"] * hw / (ti["median_ms"] * 1e-3) / 1e9
    er["gather_GB_per_s"] = er["gathered_bytes_per_pixel"] * hw / (tr["median_ms"] * 1e-3) / 1e9
    out.append(_row("mast3r_slam_backends.iter_proj 1x384x512", ti, 65 * hw, **ei))
    out.append(_row("mast3r_slam_backends.refine_matches 1x384x512 fp16", tr, 128 * hw, **er))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--full-cpu", action="store_true", help="also time the reference's host build of simple_knn.cu at P = 10^6 (about six minutes)")
    ap.add_argument("--only", default=None, choices=[None, "knn", "matching"])
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    rows = []
    if a.only in (None, "knn"):
        rows += knn_rows(dev, a.reps, not a.no_cpu, a.full_cpu)
    if a.only in (None, "matching"):
        rows += matching_rows(dev, a.reps, not a.no_cpu)
    for r in rows:
        print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Per-stage roofline of one optimisation step, "by formula" next to "by counters" (round 4).

    python tools/stage_roofline_table.py <bench.json> <kernel_table.json>
bench.json: the default bench line (its `roofline_stages`: SURVEY 8(d)'s algorithmic bytes / the stage's HIP-event time);
kernel_table.json: tools/kernel_table.py --json (per-kernel mean time and 2 x FETCH_SIZE + WRITE_SIZE of the SAME workload under rocprofv3,
warm-up-scene launches dropped).  A stage's counter bytes / time = the sum over the kernels it launches."""
import json
import sys

STAGE_KERNELS = {
    "project_fwd": ["adk::project_fwd_kernel"], "lod_project_fwd": ["adk::lod_project_fwd_kernel"],
    "binning": ["bin_count_kernel", "bin_colscan_kernel", "bin_tilescan_kernel", "bin_scatter_kernel", "bin_tile_sort_"],
    "raster_fwd": ["raster_fwd_kernel"], "raster_bwd": ["raster_bwd_kernel"], "project_bwd": ["project_bwd_kernel"],
    "ssim_fwd": ["ssim_fwd_kernel"], "ssim_bwd": ["ssim_bwd_kernel"], "adam_multi": ["adam_multi_kernel"],
    "lod_params_fwd": ["adk::lod_params_fwd_kernel"], "lod_params_bwd": ["lod_params_bwd_kernel", "lod_reduce_partials_kernel"],
    "photometric_fwd": ["photometric_fwd_kernel"], "photometric_bwd": ["photometric_bwd_kernel"],
}


def main():
    bench = json.loads([ln for ln in open(sys.argv[1]) if ln.startswith("{")][-1])
    table = json.load(open(sys.argv[2]))
    rs = bench["roofline_stages"]
    # calls per optimisation step: normalise by the kernel that runs exactly once per step
    steps = max(table[k]["calls"] for k in table if "adam_multi_kernel" in k)
    print(f"# {sys.argv[1]} (HIP events, formula bytes) vs {sys.argv[2]} (rocprofv3, counter bytes; {steps} optimisation steps)")
    print(f"{'stage':16s} {'formula MB':>10s} {'event ms':>9s} {'frac':>6s} | {'counter MB':>10s} {'rocprof ms':>10s} {'frac':>6s} | counter / formula")
    for st, pats in STAGE_KERNELS.items():
        ks = [k for k in table if any(p in k for p in pats)]
        if st not in rs or not ks:
            continue
        n = max(table[k]["calls"] for k in ks)   # per launch of the stage: its most-launched kernel (test keyframes run no Adam, the first frames no sort)
        us = sum(table[k]["mean_us"] * table[k]["calls"] for k in ks) / n
        mb = sum(table[k].get("hbm_mb", 0.0) * table[k]["calls"] for k in ks) / n
        f = rs[st]
        frac_c = (mb / us) * 1000 / 8000.0 if us > 0 else 0.0
        print(f"{st:16s} {f['alg_bytes'] / 1e6:10.1f} {f['ms']:9.4f} {f['frac']:6.3f} | {mb:10.1f} {us / 1e3:10.4f} {frac_c:6.3f} | {mb / (f['alg_bytes'] / 1e6):5.2f}")


if __name__ == "__main__":
    main()

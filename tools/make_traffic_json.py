#!/usr/bin/env python
"""profiles/traffic.json from the three rocprofv3 --pmc passes of tools/profile_round.sh (FETCH_SIZE, WRITE_SIZE, SQ_*),
stamped with the hash of the kernel sources they were measured on (bench.kernel_source_sha): bench.py reports
`roofline.traffic` / `roofline_valu` from this file ONLY while raster_tiles.hip is unchanged.

    python tools/make_traffic_json.py <fetch_dir> <write_dir> <sq_dir> <tag> <gaussians> <width> <height> > profiles/traffic.json
HBM bytes per launch = 2 x FETCH_SIZE (gfx950: FETCH_SIZE tallies 128 B requests of wide coalesced reads at 64 B,
MI355X_MICROARCH.md "HBM") + WRITE_SIZE, both reported by rocprofv3 in KB, separate passes."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from pmc_agg import agg  # noqa: E402
import bench  # noqa: E402

f, w, s = (agg(p + "/b_counter_collection.csv") for p in sys.argv[1:4])
tag, n, wd, ht = sys.argv[4], int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])
# the form the bench frame launches = the raster_bwd instantiation with the most wave cycles (the warm-up scene's 96x64 frames use the quadrant form)
k = max((k for k in s if "raster_bwd_kernel<" in k), key=lambda k: s[k].get("SQ_WAVE_CYCLES", 0.0))
fetch_kb, write_kb = f[k]["FETCH_SIZE"], w[k]["WRITE_SIZE"]
out = {"tag": tag, "kernel": k, "gaussians": n, "width": wd, "height": ht, "kernel_source_sha": bench.kernel_source_sha(),
       "raster_bwd_hbm_bytes": (2.0 * fetch_kb + write_kb) * 1024.0,
       "raster_bwd_fetch_size_kb": fetch_kb, "raster_bwd_write_size_kb": write_kb,
       "raster_bwd_valu_wave_insts": s[k].get("SQ_INSTS_VALU"),
       "raster_bwd_active_inst_valu_quadcycles": s[k].get("SQ_ACTIVE_INST_VALU"),
       "raster_bwd_salu_insts": s[k].get("SQ_INSTS_SALU"), "raster_bwd_lds_insts": s[k].get("SQ_INSTS_LDS"),
       "raster_bwd_wave_cycles_quad": s[k].get("SQ_WAVE_CYCLES"), "raster_bwd_waves": s[k].get("SQ_WAVES"),
       "note": "per-launch means over the launches of one bench run; separate --pmc passes; 2 x FETCH_SIZE + WRITE_SIZE"}
print(json.dumps(out, indent=1))

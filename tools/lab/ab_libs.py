#!/usr/bin/env python
"""Same-box A/B of library builds (lib/libartdeco_hip.<variant>.so from tools/lab/ab_step.py --build ...) on the stationary
optimisation step: runs tools/lab/ab_tile_shape.py per library, interleaved twice.   python tools/lab/ab_libs.py [N W H] [--shapes=16x16]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIBDIR = os.path.join(ROOT, "artdeco_amd", "lib")
libs = {"default": os.path.join(LIBDIR, "libartdeco_hip.so")}
for f in sorted(os.listdir(LIBDIR)):
    if f.startswith("libartdeco_hip.") and f.endswith(".so") and f != "libartdeco_hip.so":
        libs[f[len("libartdeco_hip."):-3]] = os.path.join(LIBDIR, f)
args = sys.argv[1:] or ["1000000", "1920", "1080", "--shapes=16x16"]
for rep in range(2):
    for name, so in libs.items():
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "lab", "ab_tile_shape.py"), *args], env=dict(os.environ, ARTDECO_HIP_LIB=so),
                           capture_output=True, text=True)
        try:
            d = json.loads(r.stdout[r.stdout.index("{"):])
        except Exception:
            print(name, "FAILED", r.stderr[-400:])
            continue
        for cfg, res in d.items():
            for shape, runs in res.items():
                print(f"{name:10s} {cfg} {shape} " + " | ".join(f"fwd {x['raster_fwd']} bwd {x['raster_bwd']} step {x['step_ms']}" for x in runs), flush=True)

#!/usr/bin/env python
"""Same-box A/B of the internal tile shape (ADK_TILE_SHAPE = 16x16 | 32x16) on the stationary optimisation step: per-stage HIP-event
times of both shapes, alternating in one process.   python tools/lab/ab_tile_shape.py [N W H]..."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

import artdeco_amd

artdeco_amd.install_dropins()
from artdeco_amd import fused, rasterizer
from harness import mapper

dev = torch.device("cuda:0")
cfgs = [(1_000_000, 1920, 1080), (1_000_000, 512, 384), (1_000_000, 648, 486), (4_000_000, 2592, 1944), (200_000, 512, 384)]
SHAPES = ("16x16", "32x16")
for a in list(sys.argv):
    if a.startswith("--shapes="):
        SHAPES = tuple(a.split("=", 1)[1].split(","))
        sys.argv.remove(a)
if len(sys.argv) > 3:
    cfgs = [tuple(int(x) for x in sys.argv[1:4])]
out = {}
for N, W, H in cfgs:
    scene = mapper.build_synthetic_mapper(N, W, H, dev, seed=0, targets="render")
    fused.patch_scene_model(scene)
    res = {}
    for rep in range(2):
        for shape in SHAPES:
            os.environ["ADK_TILE_SHAPE"] = shape
            for i in range(5):
                scene.optimization_step(i % 4)
            t = rasterizer.StageTimer()
            rasterizer.set_stage_timer(t)
            for i in range(20):
                scene.optimization_step(i % 4)
            rasterizer.set_stage_timer(None)
            sm = t.summary_ms()
            torch.cuda.synchronize()
            import time
            t0 = time.perf_counter()
            for i in range(30):
                scene.optimization_step(i % 4)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 30 * 1e3
            r = {k: round(v["mean_ms"], 4) for k, v in sm.items() if k in ("bin_count", "bin_scatter", "bin_sort", "raster_fwd", "raster_bwd", "project_bwd")}
            r["step_ms"] = round(ms, 4)
            r["I"] = rasterizer.LAST_STATS["I"]
            res.setdefault(shape, []).append(r)
    out[f"{N}/{W}x{H}"] = res
    del scene
    torch.cuda.empty_cache()
print(json.dumps(out, indent=1))

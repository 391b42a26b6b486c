#!/usr/bin/env python
"""How much of the optimisation step is the GPU when the map is small (the first few hundred frames of every run)?
ms per step (wall, 60 steps) next to the sum of the stage times (HIP events) at 512x384.   python tools/lab/small_map_lab.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

import artdeco_amd

artdeco_amd.install_dropins()
from artdeco_amd import fused, rasterizer
from harness import mapper

dev = torch.device("cuda:0")
out = {}
NS = [int(x) for x in sys.argv[1:]] or [5_000, 20_000, 50_000, 100_000, 200_000, 500_000]
for N in NS:
    scene = mapper.build_synthetic_mapper(N, 512, 384, dev, seed=0, targets="render")
    fused.patch_scene_model(scene)
    fused.freeze_gc()
    for i in range(10):
        scene.optimization_step(i % 4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(60):
        scene.optimization_step(i % 4)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 60 * 1e3
    t = rasterizer.StageTimer()
    rasterizer.set_stage_timer(t)
    for i in range(20):
        scene.optimization_step(i % 4)
    rasterizer.set_stage_timer(None)
    busy = sum(v["mean_ms"] for v in t.summary_ms().values())
    out[N] = {"ms_per_step": round(wall, 4), "sum_of_stage_ms": round(busy, 4), "I": rasterizer.LAST_STATS["I"]}
    print(N, out[N], flush=True)
    del scene
    torch.cuda.empty_cache()

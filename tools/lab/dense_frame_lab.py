#!/usr/bin/env python
"""DEV TOOL (round 5): the optimisation step on a DENSE north-star-size frame (1 M Gaussians, 512x384, sigma ~ 4 px: every tile list 10-12 k
entries) -- the long-list sort inside the one-call step against round 4's way (per-stage chain + global radix route), with the binning stages'
HIP-event times.    python tools/lab/dense_frame_lab.py [sigma_px]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

import artdeco_amd

artdeco_amd.install_dropins()
from artdeco_amd import fused, native_step, rasterizer
from harness import mapper

sigma = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
dev = torch.device("cuda:0")
scene = mapper.build_synthetic_mapper(1_000_000, 512, 384, dev, seed=0, targets="render", sigma_px=sigma)
fused.patch_scene_model(scene)
for mode, env in (("native + long-list sort", {}), ("round-4 path: per-stage chain + global radix route", {"ARTDECO_AMD_NATIVE_STEP": "0", "ADK_BIN_LONG": "0"}),
                  ("per-stage chain + long-list sort", {"ARTDECO_AMD_NATIVE_STEP": "0"})):
    os.environ.update(env)
    for rep in range(2):
        for i in range(5):
            scene.optimization_step(i % 4)
        t = rasterizer.StageTimer()
        rasterizer.set_stage_timer(t)
        for i in range(20):
            scene.optimization_step(i % 4)
        sm = t.summary_ms()
        rasterizer.set_stage_timer(None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(30):
            scene.optimization_step(i % 4)
        torch.cuda.synchronize()
        step = (time.perf_counter() - t0) / 30 * 1e3
        print(f"sigma {sigma} px, I = {rasterizer.LAST_STATS.get('I')}: {mode:52s} step {step:.3f} ms | " +
              " ".join(f"{k} {sm[k]['mean_ms']:.3f}" for k in ("bin_count", "bin_scatter", "bin_sort", "raster_fwd", "raster_bwd") if k in sm) +
              f" | long-list steps so far {native_step.STATS['long_list_steps']}", flush=True)
    for k in env:
        del os.environ[k]

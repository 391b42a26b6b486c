#!/usr/bin/env python
"""Mean HIP-event time of chosen stages of the stationary optimisation step (for A/Bs of library variants via ARTDECO_HIP_LIB).
    python tools/lab/stage_times.py N W H stage[,stage...]"""
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

N, W, H = (int(x) for x in sys.argv[1:4])
want = sys.argv[4].split(",")
dev = torch.device("cuda:0")
scene = mapper.build_synthetic_mapper(N, W, H, dev, seed=0, targets="render")
fused.patch_scene_model(scene)
res = []
for rep in range(3):
    for i in range(5):
        scene.optimization_step(i % 4)
    t = rasterizer.StageTimer()
    rasterizer.set_stage_timer(t)
    for i in range(30):
        scene.optimization_step(i % 4)
    rasterizer.set_stage_timer(None)
    sm = t.summary_ms()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(30):
        scene.optimization_step(i % 4)
    torch.cuda.synchronize()
    res.append({**{k: sm[k]["mean_ms"] for k in want}, "step": (time.perf_counter() - t0) / 30 * 1e3})
print(os.path.basename(os.environ.get("ARTDECO_HIP_LIB", "default")), f"{N}/{W}x{H}", " | ".join(" ".join(f"{k} {v:.4f}" for k, v in r.items()) for r in res), flush=True)

#!/usr/bin/env python
"""Host cost of one optimisation step: a scene so small that the GPU is never the bound (2 000 Gaussians, 96x64), wall time per step with the
one-call native step and with the per-stage chain.    python tools/lab/host_cost_step.py [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

import artdeco_amd

artdeco_amd.install_dropins()
from artdeco_amd import fused, native_step
from harness import mapper

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
dev = torch.device("cuda:0")
scene = mapper.build_synthetic_mapper(2000, 96, 64, dev, seed=0, targets="render")
fused.patch_scene_model(scene)
fused.freeze_gc()
for rep in range(2):
    for mode in ("1", "0"):
        os.environ["ARTDECO_AMD_NATIVE_STEP"] = mode
        for i in range(50):
            scene.optimization_step(i % 4)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            scene.optimization_step(i % 4)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps * 1e3
        print(f"native={mode} host-bound step {dt:.4f} ms  ({native_step.STATS})", flush=True)

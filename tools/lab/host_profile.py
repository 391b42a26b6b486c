#!/usr/bin/env python
"""DEV TOOL: where does the HOST time of one fused mapper step go?  cProfile over N steps of the bench scene at a size where the
GPU is faster than the Python that feeds it (default 1 M Gaussians, 512x384)."""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import artdeco_amd
artdeco_amd.install_dropins()
from artdeco_amd import fused
from harness import mapper
N, W, H = (int(a) for a in (sys.argv[1:4] if len(sys.argv) >= 4 else (1_000_000, 512, 384)))
dev = torch.device("cuda:0")
scene = mapper.build_synthetic_mapper(N, W, H, dev, seed=0, targets="render")
fused.patch_scene_model(scene)
nkf = len(scene.keyframes)
for i in range(20):
    scene.optimization_step(i % nkf)
torch.cuda.synchronize()
steps = 300
t0 = time.perf_counter()
for i in range(steps):
    scene.optimization_step(i % nkf)
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / steps * 1e3:.3f} ms/step unprofiled")
# host-only cost: how long does enqueueing take when nothing waits for the GPU?  (the step reads one count back, so this is an upper bound)
pr = cProfile.Profile()
pr.enable()
for i in range(steps):
    scene.optimization_step(i % nkf)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(35)
print(s.getvalue()[:6000])

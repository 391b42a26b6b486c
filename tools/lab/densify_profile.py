#!/usr/bin/env python
"""One add_new_gaussians call per iteration on the headline scene (1 M Gaussians, 1920x1080), for `rocprofv3 --kernel-trace --stats`:
which kernels an important frame's densification spends its time in.   python tools/lab/densify_profile.py [N W H iters]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

import artdeco_amd

artdeco_amd.install_dropins()
from artdeco_amd import fused
from harness import mapper, stream

a = [int(x) for x in sys.argv[1:]]
N, W, H, IT = (a + [1_000_000, 1920, 1080, 6][len(a):])[:4]
dev = torch.device("cuda:0")
scene = mapper.build_synthetic_mapper(N, W, H, dev, seed=0, n_keyframes=0, targets="random")
fused.patch_scene_model(scene)
frames = stream.synthetic_frames(scene, IT + 2, seed=0, texture=0.05)
ts = []
for i, fr in enumerate(frames):
    scene.add_keyframe(stream.make_keyframe(scene, fr, index=i))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    scene.add_new_gaussians()
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print("add_new_gaussians ms per call:", [round(t, 2) for t in ts], "N =", scene.xyz.shape[0])

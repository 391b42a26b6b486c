#!/usr/bin/env python
"""Wall time of every frame of the mapper loop on a fresh scene (one device synchronisation per frame): where one-off costs land.
    python tools/lab/frame_series.py N MAP_W MAP_H [PYR_LEVELS] [FRAMES]
BATCHED=1: the SLAM-keyframe pose re-read through artdeco_amd.keyframe_poses.update_keyframe_poses instead of run_system.py's per-keyframe loop."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import artdeco_amd

artdeco_amd.install_dropins()
from artdeco_amd import fused
from harness import mapper, stream

n, w, h = (int(x) for x in sys.argv[1:4])
pyr = int(sys.argv[4]) if len(sys.argv) > 4 else 1
nf = int(sys.argv[5]) if len(sys.argv) > 5 else 48
dev = torch.device("cuda:0")
scene = mapper.build_synthetic_mapper(n, w, h, dev, seed=0, n_keyframes=0, targets="random")
fused.patch_scene_model(scene)
frames = stream.synthetic_frames(scene, nf, seed=0, texture=0.05)
np.random.seed(0)
stream.warm_process(dev) if os.environ.get('WARM', '1') == '1' else stream.warm_libraries(dev)
clock = stream.StageClock(False)
for i, fr in enumerate(frames):
    fl = stream.frame_flags(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps = stream.run_frame(scene, fr, len(scene.keyframes), fl, clock, pyr_levels=pyr, batched_slam_update=os.environ.get('BATCHED', '0') == '1')
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    tag = ("S" if fl["is_slam_keyframe"] and i > 0 else "-") + ("I" if fl["is_important"] else "-") + ("T" if fl["is_test"] else "-")
    print(f"frame {i:3d} {tag} steps {steps:2d}  {ms:8.2f} ms  {ms / steps:6.3f} ms/step  N={scene.xyz.shape[0]}", flush=True)

#!/usr/bin/env python
"""DEV TOOL (round 6): where the host time of run_system.py's SLAM-keyframe loop (harness/stream.slam_pose_update, :194-227 as written) goes at K
keyframes -- wall per keyframe, device-busy fraction, and a cProfile of the Python side.   python tools/lab/slam_loop_profile.py [K=500]"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

import artdeco_amd

artdeco_amd.install_dropins()
from artdeco_amd import fused, small_inverse
from harness import mapper, stream

K = int(sys.argv[1]) if len(sys.argv) > 1 else 500
dev = torch.device("cuda:0")
scene = mapper.build_synthetic_mapper(200_000, 512, 384, dev, seed=0, n_keyframes=0, targets="random")
fused.patch_scene_model(scene)
frames = stream.synthetic_frames(scene, 8, seed=0)
stream.warm_process(dev)
stream.fast_forward(scene, frames, K)
for rep in range(2):
    stream.slam_pose_update(scene, seed=rep)
torch.cuda.synchronize()
t0 = time.perf_counter()
for rep in range(3):
    stream.slam_pose_update(scene, seed=10 + rep)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print(f"K = {K}: {dt * 1e3:.1f} ms per SLAM keyframe = {dt / K * 1e6:.1f} us per keyframe; inverse stats {small_inverse.STATS}", flush=True)
pr = cProfile.Profile()
pr.enable()
stream.slam_pose_update(scene, seed=99)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr, stream=sys.stdout)
st.sort_stats("cumulative").print_stats(28)

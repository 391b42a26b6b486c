#!/usr/bin/env python
"""DEV TOOL (round 5): bench.py's `full_sequence` for ONE configuration, e.g. to A/B ARTDECO_AMD_FUSE_POSE (Keyframe.get_Rt / set_Rt as one
launch each) on the reference script's per-keyframe SLAM loop.    python tools/lab/full_sequence_lab.py N W H FRAMES [batched]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

import bench

n, w, h, nf = (int(x) for x in sys.argv[1:5])
batched = len(sys.argv) > 5 and sys.argv[5] == "batched"
args = argparse.Namespace(kf_every=5, slam_every=15, test_hold=8, texture=0.05)
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
r = bench._full_sequence(n, w, h, dev, args, nf, batched)
print(json.dumps({"fuse_pose": os.environ.get("ARTDECO_AMD_FUSE_POSE", "1"), "frames_per_s": round(r["frames_per_s"], 2),
                  "per_100": r["frames_per_s_per_100_frames"], "last_20": round(r["frames_per_s_last_20_frames"], 2), "seconds": round(r["seconds"], 2),
                  "slam_pose_update": r["slam_pose_update"]}))

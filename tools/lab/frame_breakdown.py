#!/usr/bin/env python
"""Per-stage wall time of the mapper's frame loop (harness/stream.py, synchronised per stage) at a configuration:
    python tools/lab/frame_breakdown.py N MAP_W MAP_H [PYR_LEVELS]      e.g. 1000000 1296 972 2  (run.sh geometry)"""
import json
import os
import sys

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
dev = torch.device("cuda:0")
scene = mapper.build_synthetic_mapper(n, w, h, dev, seed=0, n_keyframes=0, targets="random")
fused.patch_scene_model(scene)
cad = dict(kf_every=5, slam_every=15, test_hold=8)
frames = stream.synthetic_frames(scene, 6 + 16 + 16, seed=0, texture=0.05)
np.random.seed(0)
stream.warm_process(dev)
stream.run_stream(scene, frames[:6], start_index=0, pyr_levels=pyr, **cad)
r = stream.run_stream(scene, frames[6:22], start_index=6, pyr_levels=pyr, **cad)
b = stream.run_stream(scene, frames[22:], start_index=22, pyr_levels=pyr, breakdown=True, **cad)
print(json.dumps({"config": [n, w, h, pyr], "frames_per_s": r["frames"] / r["seconds"], "ms_per_frame": 1e3 * r["seconds"] / r["frames"],
                  "steps": r["steps"], "added": r["gaussians_added"],
                  "stage_ms": {k: {kk: round(vv, 3) for kk, vv in v.items()} for k, v in b["stage_ms"].items()}}, indent=1))

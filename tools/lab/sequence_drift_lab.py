#!/usr/bin/env python
"""DEV TOOL (round 6): why is the middle of a 1 000-frame run slower than a fast-forwarded scene at the same position?  Runs configs[2]'s sequence
from frame 0 and prints, per 50 frames: frames/s, the map size, the stage breakdown (device-synchronised: slower in absolute terms), torch's
allocator counters.    python tools/lab/sequence_drift_lab.py [frames=600]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import artdeco_amd

artdeco_amd.install_dropins()
from artdeco_amd import fused, native_step, rasterizer
from harness import mapper, stream

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 600
breakdown = "--breakdown" in sys.argv
stages = "--stages" in sys.argv
dev = torch.device("cuda:0")
NG, WW, HH = (int(x) for x in os.environ.get("DRIFT_SCENE", "1000000,1920,1080").split(","))
scene = mapper.build_synthetic_mapper(NG, WW, HH, dev, seed=0, n_keyframes=0, targets="random")
fused.patch_scene_model(scene)
cadence = dict(kf_every=5, slam_every=15, test_hold=8)
base = stream.synthetic_frames(scene, 48, seed=0, texture=0.05)
np.random.seed(0)
stream.warm_process(dev)
CH = 50
for c0 in range(0, n_frames, CH):
    frames = [base[j % len(base)] for j in range(c0, c0 + CH)]
    st0 = dict(native_step.STATS)
    m0 = torch.cuda.memory_stats(dev)
    if stages:
        tm = rasterizer.StageTimer()
        rasterizer.set_stage_timer(tm)
    r = stream.run_stream(scene, frames, start_index=c0, breakdown=breakdown, **cadence)
    if stages:
        rasterizer.set_stage_timer(None)
        sm = tm.summary_ms()
    m1 = torch.cuda.memory_stats(dev)
    line = (f"frames {c0:4d}-{c0 + CH:4d}: {r['frames'] / r['seconds']:6.2f} frames/s  N {r['gaussians_start']:8d} -> {r['gaussians_end']:8d} (+{r['gaussians_added']})  "
            f"steps {r['steps']}  mallocs {m1['num_device_alloc'] - m0['num_device_alloc']} frees {m1['num_device_free'] - m0['num_device_free']} "
            f"reserved {m1['reserved_bytes.all.current'] / 2**30:.1f} GiB  plans_built {native_step.STATS['plans_built'] - st0['plans_built']} "
            f"cap_retries {native_step.STATS['capacity_retries'] - st0['capacity_retries']} fallback {native_step.STATS['fallback_route'] - st0['fallback_route']}/{native_step.STATS['fallback_layout'] - st0['fallback_layout']}")
    if breakdown:
        line += "  " + " ".join(f"{k} {v['ms_per_frame']:.2f}" for k, v in r["stage_ms"].items())
    if stages:
        line += f"  I {rasterizer.LAST_STATS.get('I')}  " + " ".join(f"{k} {v['mean_ms']:.3f}" for k, v in sm.items() if v['mean_ms'] > 0.02)
    print(line, flush=True)

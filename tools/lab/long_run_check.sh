#!/bin/bash
# 300 frames of the north-star stream with the one-call step (SLAM-keyframe loop batched so that the step dominates): per-frame times,
# the native step's counters and the allocator's footprint at the end
set -u
mkdir -p gpurun_out
BATCHED=1 timeout 900 python - <<'PY' > gpurun_out/r04_long_run_300.txt 2>&1
import os, sys, time
sys.argv = ["frame_series", "1000000", "512", "384", "1", "300"]
sys.path.insert(0, os.getcwd())
exec(open("tools/lab/frame_series.py").read())
import torch
from artdeco_amd import native_step
print("native_step.STATS", native_step.STATS)
print("torch allocated GB", round(torch.cuda.memory_allocated() / 2**30, 2), "reserved GB", round(torch.cuda.memory_reserved() / 2**30, 2))
PY
python - <<'PY'
import re
rows = [l for l in open("gpurun_out/r04_long_run_300.txt") if l.startswith("frame")]
ms = [float(re.search(r"([\d.]+) ms  ", l).group(1)) for l in rows]
for a in range(0, len(ms), 50):
    w = ms[a:a + 50]
    print(f"frames {a:3d}-{a + len(w) - 1:3d}: {1e3 * len(w) / sum(w):6.2f} frames/s   max frame {max(w):7.2f} ms")
print(open("gpurun_out/r04_long_run_300.txt").read().splitlines()[-2:])
PY

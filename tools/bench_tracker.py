#!/usr/bin/env python
"""Frontend tracker alone: `python tools/bench_tracker.py [--iters 50] [--cpu-baseline]` prints one JSON
line (eager launches and one-hipGraph replay).  bench_frontend.py reports the same numbers next to the MASt3R pair match."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench_frontend  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--cpu-baseline", action="store_true")
    a = ap.parse_args()
    print(json.dumps(bench_frontend.tracker_bench(torch.device("cuda:0"), a.iters, a.cpu_baseline)))

#!/usr/bin/env python
"""DEV TOOL: per-launch mean of every counter of one `rocprofv3 --pmc ... --output-format csv` directory, for kernels whose name
contains a filter; launches whose grid is below `min_grid` workgroups (the warm-up scene) are dropped.
    python tools/lab/pmc_one.py <dir> <name filter> [min_grid]"""
import collections
import csv
import glob
import sys

d, filt = sys.argv[1], sys.argv[2]
min_grid = int(sys.argv[3]) if len(sys.argv) > 3 else 0
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if filt not in k:
            continue
        grid = int(r.get("Grid_Size", "0") or 0) // max(int(r.get("Workgroup_Size", "1") or 1), 1)
        if grid < min_grid:
            continue
        short = k.split("(")[0].replace("void ", "")
        acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k, " ".join(f"{n}={sum(v) / len(v) / 1e6:.3f}M[{len(v)}]" for n, v in sorted(acc[k].items())))

#!/usr/bin/env python
"""Compact table of a rocprofv3 *_kernel_stats.csv: calls, mean us, total ms, share -- kernel names cut at the first '('.
    python tools/kernel_stats_table.py stats.csv [top]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"{'kernel':58s} {'calls':>6s} {'mean us':>9s} {'total ms':>9s} {'share':>6s}")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:top]:
    name = r["Name"].replace("void ", "").split("(")[0][:58]
    print(f"{name:58s} {int(r['Calls']):6d} {float(r['AverageNs']) / 1e3:9.1f} {float(r['TotalDurationNs']) / 1e6:9.2f} {100 * float(r['TotalDurationNs']) / tot:5.1f}%")
print(f"{'all kernels':58s} {sum(int(r['Calls']) for r in rows):6d} {'':9s} {tot / 1e6:9.2f}")

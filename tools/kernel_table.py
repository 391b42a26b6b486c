#!/usr/bin/env python
"""Per-kernel table of one profiled bench run WITHOUT the warm-up scene (round 4).

`stream.warm_process` runs nine frames of a 2 000-Gaussian / 96x64 throw-away scene before anything is timed; rocprofv3's own
`--stats` means fold those tiny launches in (e.g. project_bwd 206 us "mean" against 257 us by events).  This tool reads the raw
per-dispatch rows instead and drops, per kernel name, every launch whose grid is below a quarter of the largest grid that kernel
was launched with in the run -- the warm-up scene's launches -- so that time AND counters are per-launch means of the bench scene.

    python tools/kernel_table.py <kernel_trace.csv> [<fetch_dir> <write_dir>] [--json out.json]
      kernel_trace.csv : rocprofv3 --kernel-trace --output-format csv  (*_kernel_trace.csv)
      fetch_dir/write_dir : rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the SAME command (counter_collection.csv inside)
HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE (KB; MI355X_MICROARCH.md "HBM": gfx950 tallies 128 B read requests at 64 B)."""
import collections
import csv
import glob
import json
import sys

HBM_PEAK = 8000.0  # GB/s


def short(name):
    return name.replace("void ", "").split("(")[0]


def rows_of(path):
    return list(csv.DictReader(open(path)))


def grid_of(r):
    for gk, wk in (("Grid_Size", "Workgroup_Size"), ("Grid_Size_X", "Workgroup_Size_X")):
        if gk in r and r[gk] not in ("", None):
            try:
                return int(r[gk]) // max(int(r.get(wk, 1) or 1), 1)
            except ValueError:
                pass
    return 0


def keep_filter(rows, name_key):
    mx = collections.Counter()
    for r in rows:
        k = short(r[name_key])
        mx[k] = max(mx[k], grid_of(r))
    return lambda r: grid_of(r) * 4 >= mx[short(r[name_key])]


def counters(d, counter):
    paths = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(list)
    for p in paths:
        rows = rows_of(p)
        keep = keep_filter(rows, "Kernel_Name")
        for r in rows:
            if r["Counter_Name"] == counter and keep(r):
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    trace = rows_of(args[0])
    keep = keep_filter(trace, "Kernel_Name")
    dur = collections.defaultdict(list)
    dropped = collections.Counter()
    for r in trace:
        k = short(r["Kernel_Name"])
        if "adk::" not in k:
            continue
        if keep(r):
            dur[k].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
        else:
            dropped[k] += 1
    fetch = counters(args[1], "FETCH_SIZE") if len(args) > 2 else {}
    write = counters(args[2], "WRITE_SIZE") if len(args) > 2 else {}
    tot = sum(sum(v) for v in dur.values())
    out = {}
    print(f"# launches of the bench scene only ({sum(dropped.values())} warm-up-scene launches dropped: grid < 1/4 of the kernel's largest grid)")
    print(f"{'kernel':52s} {'calls':>6s} {'mean us':>9s} {'total ms':>9s} {'share':>6s} {'FETCH KB':>10s} {'WRITE KB':>10s} {'HBM MB':>8s} {'TB/s':>6s} {'of 8 TB/s':>9s}")
    for k in sorted(dur, key=lambda k: -sum(dur[k])):
        v = dur[k]
        mean = sum(v) / len(v)
        line = f"{k[:52]:52s} {len(v):6d} {mean:9.1f} {sum(v) / 1e3:9.2f} {100 * sum(v) / tot:5.1f}%"
        rec = {"calls": len(v), "mean_us": mean, "dropped_warm_launches": dropped.get(k, 0)}
        if k in fetch or k in write:
            f, w = fetch.get(k, 0.0), write.get(k, 0.0)
            mb = (2 * f + w) * 1024 / 1e6
            tbs = mb / mean / 1e6 * 1e6 / 1e6 if mean > 0 else 0.0   # MB / us = TB/s
            tbs = mb / mean
            line += f" {f:10.0f} {w:10.0f} {mb:8.1f} {tbs:6.2f} {tbs * 1000 / HBM_PEAK:9.3f}"
            rec.update(fetch_kb=f, write_kb=w, hbm_mb=mb, tb_per_s=tbs, frac_of_8tbs_by_counters=tbs * 1000 / HBM_PEAK)
        out[k] = rec
        print(line)
    for a in sys.argv[1:]:
        if a.startswith("--json="):
            json.dump(out, open(a.split("=", 1)[1], "w"), indent=1)


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Kernel stats / one-frame timeline from a rocprofv3 results .db (ROCm 7 writes rocpd sqlite instead of CSV).
usage: rocpd_timeline.py results.db stats            -> CSV of per-kernel calls / mean / total / min / max (ns)
       rocpd_timeline.py results.db frame <marker-kernel-substring> <launches>   -> timeline of the last window that
       starts at a kernel whose name contains the marker and spans exactly <launches> launches"""
import sqlite3
import sys


def tables(c):
    names = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    return [t for t in names if "kernel_dispatch" in t][0], [t for t in names if "kernel_symbol" in t][0]


def main():
    c = sqlite3.connect(sys.argv[1])
    kd, ks = tables(c)
    if sys.argv[2] == "stats":
        q = (f"select s.kernel_name, count(*), avg(d.end-d.start), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
             f"from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 4 desc")
        print('"Name","Calls","AverageNs","TotalDurationNs","MinNs","MaxNs"')
        for r in c.execute(q):
            print('"%s",%d,%.1f,%d,%d,%d' % r)
        return
    marker, span = sys.argv[3], int(sys.argv[4])
    rows = list(c.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
    idx = [i for i, r in enumerate(rows) if marker in r[0]]
    wins = [j for j in range(len(idx) - 1) if idx[j + 1] - idx[j] == span]
    i0, i1 = idx[wins[-1]], idx[wins[-1] + 1]
    prev, busy = None, 0.0
    for name, st, en in rows[i0:i1]:
        short = name.split("adk")[-1][:48] if "adk" in name else name[:48]
        print("%-50s dur %7.1f us  gap %6.1f" % (short, (en - st) / 1e3, 0.0 if prev is None else (st - prev) / 1e3))
        prev, busy = en, busy + (en - st) / 1e3
    print("span %.1f us, busy %.1f us, launches %d" % ((rows[i1 - 1][2] - rows[i0][1]) / 1e3, busy, i1 - i0))


if __name__ == "__main__":
    main()

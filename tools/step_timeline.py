"""Print the kernel timeline of the last complete mapper step from a rocprofv3 kernel trace CSV
(`rocprofv3 --kernel-trace --output-format csv`): start offset, gap to the previous kernel, duration, name.
Steps are delimited by the adam_multi kernel that ends each one.  usage: step_timeline.py trace.csv [step_index | mid]
(default: the last step; bench.py's last 10 steps carry per-stage HIP events -- 5-10 us in front of every stage --, earlier ones only
raster_bwd's: `mid` = the step in the middle of the trace, i.e. one from INSIDE the timed region)"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "adam_multi" in r["Kernel_Name"]]
k = (len(idx) // 2 if sys.argv[2] == "mid" else int(sys.argv[2])) if len(sys.argv) > 2 else len(idx) - 1
a, b = idx[k - 1], idx[k]
t0 = int(rows[a]["End_Timestamp"])
prev = t0
busy = 0
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += e - s
    print("%9.1f gap %6.1f dur %7.1f  %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:100]))
    prev = e
print("step span %.1f us, busy %.1f us, kernels %d" % ((prev - t0) / 1e3, busy / 1e3, b - a))

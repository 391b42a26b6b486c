"""Per-launch durations (us) of the kernels whose name contains argv[2], in launch order, from a rocprofv3
kernel-trace CSV.  usage: kernel_durations.py trace.csv raster_bwd"""
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print(" ".join("%.0f" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows))

import collections
import csv
import sys


def agg(path):
    rows = list(csv.DictReader(open(path)))
    a = collections.defaultdict(lambda: collections.defaultdict(float))
    c = collections.defaultdict(collections.Counter)
    # round 4: launches of the warm-up scene (stream.warm_process: 2 000 Gaussians, 96x64) are dropped -- per kernel name, every launch whose
    # grid is below a quarter of that kernel's largest grid in the run (the same rule as tools/kernel_table.py)
    def grid(r):
        try:
            return int(r.get("Grid_Size", 0) or 0) // max(int(r.get("Workgroup_Size", 1) or 1), 1)
        except ValueError:
            return 0
    mx = collections.Counter()
    for r in rows:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        mx[k] = max(mx[k], grid(r))
    for r in rows:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "adk::" not in k or grid(r) * 4 < mx[k]:
            continue
        a[k][r["Counter_Name"]] += float(r["Counter_Value"])
        c[k][r["Counter_Name"]] += 1
    return {k: {n: v / c[k][n] for n, v in d.items()} for k, d in a.items()}


if __name__ == "__main__":
  f, w, s = (agg(p + "/b_counter_collection.csv") for p in sys.argv[1:4])
  print("# per-launch means; FETCH_SIZE/WRITE_SIZE in KB (separate --pmc passes), SQ/GRBM counters in millions")
  for k in sorted(s):
    sq = "  ".join("%s=%.1f" % (n.replace("SQ_", ""), v / 1e6) for n, v in sorted(s[k].items()))
    print("%-44s FETCH_KB=%9.0f WRITE_KB=%9.0f  %s" % (k, f.get(k, {}).get("FETCH_SIZE", 0), w.get(k, {}).get("WRITE_SIZE", 0), sq))

import collections
import csv
import sys


def agg(path):
    rows = list(csv.DictReader(open(path)))
    a = collections.defaultdict(lambda: collections.defaultdict(float))
    c = collections.defaultdict(collections.Counter)
    for r in rows:
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "adk::" not in k:
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

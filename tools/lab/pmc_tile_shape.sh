#!/bin/bash
# SQ counters of the raster kernels for both internal tile shapes (one rocprofv3 --pmc pass over tools/lab/ab_tile_shape.py).
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pts
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pts -o b -- python $ROOT/tools/lab/ab_tile_shape.py 1000000 1920 1080 > /tmp/pts.log 2>&1
python - <<PY > "$OUT/r03_pmc_tile_shape.txt"
import sys
sys.path.insert(0, "$ROOT/tools")
from pmc_agg import agg
s = agg("/tmp/pts/b_counter_collection.csv")
print("# per-launch means, millions (SQ_WAVES: count); 1 M Gaussians / 1920x1080; <2, 2> = 16x16 tiles, <4, 2> = 32x16 internal tiles")
for k in sorted(s):
    if "raster_" in k:
        print("%-40s %s" % (k, "  ".join("%s=%.2f" % (n.replace("SQ_", ""), (v / 1e6 if n != "SQ_WAVES" else v)) for n, v in sorted(s[k].items()))))
PY
tail -3 /tmp/pts.log >> "$OUT/r03_pmc_tile_shape.txt"

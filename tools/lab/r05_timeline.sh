#!/bin/bash
# round 5, late: the kernel timeline of a step from INSIDE bench.py's timed region (tools/step_timeline.py ... mid) next to the last one
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pk && timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/pk -o b -- python $ROOT/bench.py --steps 12 --warmup 6 --no-extra-configs --no-cpu-baseline --no-frontend > /tmp/pk.log 2>&1
python "$ROOT/tools/step_timeline.py" /tmp/pk/b_kernel_trace.csv mid > "$OUT/r05_step_timeline_timed_region.txt"
tail -1 "$OUT/r05_step_timeline_timed_region.txt"

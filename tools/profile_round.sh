#!/bin/bash
# Round-end evidence run on the GPU box (called through gpurun from the repo root):
#   tools/profile_round.sh <tag>      e.g.  tools/profile_round.sh r01_v4
# Writes into gpurun_out/<tag>_*: the default bench line, the rocprofv3 kernel stats of the bench step, the
# kernel timeline of one step, and three separate --pmc passes (FETCH_SIZE / WRITE_SIZE / SQ_*) folded per kernel.
# rocprofv3 is run from /tmp with --kernel-trace only next to --pmc (never with sys/hip/hsa traces).
set -u
TAG=${1:-round}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $ROOT/bench.py --steps 12 --warmup 6 --no-extra-configs --no-cpu-baseline --no-frontend"   # 12 mapped FRAMES (~160 optimisation steps)
# SKIP_BENCH=1: keep the ${TAG}_bench_full.json that is already there (only the rocprofv3 passes are repeated, e.g. after an edit of the
# roofline kernel's sources that changed no default code path but its hash)
if [ "${SKIP_BENCH:-0}" != "1" ]; then python "$ROOT/bench.py" > "$OUT/${TAG}_bench_full.json" 2> "$OUT/${TAG}_bench_full.err"; fi
rm -rf /tmp/pk && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o b -- $BENCH > /tmp/pk.log 2>&1
cp /tmp/pk/b_kernel_stats.csv "$OUT/${TAG}_step_kernel_stats.csv"
python "$ROOT/tools/step_timeline.py" /tmp/pk/b_kernel_trace.csv > "$OUT/${TAG}_step_timeline.txt"
python "$ROOT/tools/step_timeline.py" /tmp/pk/b_kernel_trace.csv mid > "$OUT/${TAG}_step_timeline_timed_region.txt"   # a step without the breakdown pass's events
rm -rf /tmp/pf /tmp/pw /tmp/ps
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf -o b -- $BENCH > /tmp/pf.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw -o b -- $BENCH > /tmp/pw.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 --kernel-trace --output-format csv -d /tmp/ps -o b -- $BENCH > /tmp/ps.log 2>&1
python "$ROOT/tools/pmc_agg.py" /tmp/pf /tmp/pw /tmp/ps > "$OUT/${TAG}_pmc.txt" 2> "$OUT/${TAG}_pmc.err"
# round 4: per-kernel table of the BENCH scene only (the warm-up scene's launches dropped by grid size), time and counter bytes side by side,
# and the per-stage roofline "by formula" next to "by counters"
python "$ROOT/tools/kernel_table.py" /tmp/pk/b_kernel_trace.csv /tmp/pf /tmp/pw --json="$OUT/${TAG}_step_kernel_table.json" > "$OUT/${TAG}_step_kernel_table.txt" 2>> "$OUT/${TAG}_pmc.err"
python "$ROOT/tools/stage_roofline_table.py" "$OUT/${TAG}_bench_full.json" "$OUT/${TAG}_step_kernel_table.json" > "$OUT/${TAG}_stage_roofline.txt" 2>> "$OUT/${TAG}_pmc.err"
# per-launch counters of the roofline kernel, stamped with the hash of its sources (bench.py reads profiles/traffic.json)
python "$ROOT/tools/make_traffic_json.py" /tmp/pf /tmp/pw /tmp/ps "$TAG" 1000000 1920 1080 > "$OUT/${TAG}_traffic.json" 2>> "$OUT/${TAG}_pmc.err"
tail -2 /tmp/ps.log > "$OUT/${TAG}_pmc_sq.log"
echo done

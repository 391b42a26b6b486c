set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for cfg in "200000 648 486" "1000000 512 384"; do
set -- $cfg
B="python $ROOT/bench.py --steps 12 --warmup 6 --no-extra-configs --no-cpu-baseline --no-frontend --gaussians $1 --width $2 --height $3"
$B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2x$3', d['value'], d['ms_per_step'], d.get('frame_stage_ms',{}).get('optimization_loop'))"
rm -rf /tmp/pk && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o b -- $B > /tmp/pk.log 2>&1
python $ROOT/tools/step_timeline.py /tmp/pk/b_kernel_trace.csv > $ROOT/gpurun_out/r03_timeline_$1_$2x$3.txt
tail -3 $ROOT/gpurun_out/r03_timeline_$1_$2x$3.txt
done

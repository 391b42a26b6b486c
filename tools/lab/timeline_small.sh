# DEV TOOL: kernel timeline of one bench step at another size:  bash tools/lab/timeline_small.sh <gaussians> <width> <height>
G=${1:-1000000}; W=${2:-512}; H=${3:-384}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pk2 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk2 -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 4 --no-extra-configs --no-cpu-baseline --no-frontend --gaussians $G --width $W --height $H > /tmp/pk2.log 2>&1
python $GRAFT_REPO_ROOT/tools/step_timeline.py /tmp/pk2/b_kernel_trace.csv > $GRAFT_REPO_ROOT/gpurun_out/timeline_${G}_${W}x${H}.txt
grep -v "dur     [0-9]\.[0-9] " $GRAFT_REPO_ROOT/gpurun_out/timeline_${G}_${W}x${H}.txt | tail -32

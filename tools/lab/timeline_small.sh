cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pk2 && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk2 -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 4 --no-extra-configs --no-cpu-baseline --no-frontend --width 512 --height 384 > /tmp/pk2.log 2>&1
python $GRAFT_REPO_ROOT/tools/step_timeline.py /tmp/pk2/b_kernel_trace.csv > $GRAFT_REPO_ROOT/gpurun_out/timeline_512.txt
tail -45 $GRAFT_REPO_ROOT/gpurun_out/timeline_512.txt

cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fe && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fe -o f -- python $GRAFT_REPO_ROOT/bench_frontend.py --iters 10 > $GRAFT_REPO_ROOT/gpurun_out/fe_bench.log 2>&1
cp /tmp/fe/f_kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/fe_kernel_stats.csv
tail -3 $GRAFT_REPO_ROOT/gpurun_out/fe_bench.log | cut -c1-600

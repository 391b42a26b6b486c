mkdir -p gpurun_out
(timeout 120 python tools/lab/reduce_lab.py > gpurun_out/r05_reduce_lab.txt 2>&1)
L=artdeco_amd/lib
(
for rep in 1 2; do
 for v in libartdeco_hip.so libartdeco_hip.rows0.so; do
  ARTDECO_HIP_LIB=$L/$v timeout 300 python tools/lab/stage_times.py 1000000 1920 1080 raster_bwd,raster_fwd 2>&1 | tail -1
  ARTDECO_HIP_LIB=$L/$v ADK_RASTER_SPLIT_BWD=0 timeout 300 python tools/lab/stage_times.py 1000000 1920 1080 raster_bwd 2>&1 | tail -1 | sed 's/^/[whole tile] /'
  ARTDECO_HIP_LIB=$L/$v timeout 300 python tools/lab/stage_times.py 1000000 512 384 raster_bwd,raster_fwd 2>&1 | tail -1
 done
done
) > gpurun_out/r05_ab_rows_first.txt 2>&1
(timeout 600 python -m pytest tests/test_raster.py -m gpu -q -x -k "backward or waves or forms" > gpurun_out/r05_raster_parity.log 2>&1; echo "rc $?" >> gpurun_out/r05_raster_parity.log)
(timeout 300 python -m pytest tests/test_ref_pinning.py -m gpu -q > gpurun_out/r05_ref_full.log 2>&1; echo "rc $?" >> gpurun_out/r05_ref_full.log)
(timeout 1200 python -m pytest tests/test_step_oracle.py -m gpu -q -s > gpurun_out/r05_step_oracle3.log 2>&1; echo "rc $?" >> gpurun_out/r05_step_oracle3.log)
cat gpurun_out/r05_reduce_lab.txt | tail -6; cat gpurun_out/r05_ab_rows_first.txt; tail -3 gpurun_out/r05_raster_parity.log; tail -3 gpurun_out/r05_ref_full.log; grep -E "step-oracle.*(knife rows|oracle [0-9])|passed|failed" gpurun_out/r05_step_oracle3.log | cut -c1-700

#!/bin/bash
# native one-call step vs per-stage chain: parity tests, then the stationary step at three sizes (same box, alternating)
set -u
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_native_step.py tests/test_fused_glue.py tests/test_raster.py -x -q -m gpu > gpurun_out/r04_native_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r04_native_tests.log
tail -5 gpurun_out/r04_native_tests.log
: > gpurun_out/r04_ab_native_step.txt
for rep in 1 2; do
for v in 1 0; do
  for cfg in "1000000 512 384" "1000000 1920 1080" "200000 512 384"; do
    ARTDECO_AMD_NATIVE_STEP=$v timeout 300 python tools/lab/stage_times.py $cfg raster_bwd 2>&1 | tail -1 | sed "s/^/native=$v /" >> gpurun_out/r04_ab_native_step.txt
  done
done
done
cat gpurun_out/r04_ab_native_step.txt

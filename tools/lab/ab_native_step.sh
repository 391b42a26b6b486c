set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_native_step.py -x -q -m gpu > gpurun_out/r04_native_tests.log 2>&1; echo "native tests rc=$?" >> gpurun_out/r04_native_tests.log
tail -15 gpurun_out/r04_native_tests.log
for v in 1 0; do
  for cfg in "1000000 512 384" "1000000 1920 1080" "200000 512 384"; do
    ARTDECO_AMD_NATIVE_STEP=$v timeout 300 python tools/lab/stage_times.py $cfg raster_bwd 2>&1 | tail -1 | sed "s/^/native=$v /" >> gpurun_out/r04_ab_native_step.txt
  done
done
cat gpurun_out/r04_ab_native_step.txt

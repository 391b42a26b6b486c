#!/bin/bash
# lod_params_bwd with / without the late request of the next chunk's stage-0 inputs (same box, alternating)
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_fused_glue.py tests/test_native_step.py -x -q -m gpu 2>&1 | tail -2 > gpurun_out/r04_ab_lod_late2.txt
for rep in 1 2; do
  for lib in default late1; do
    for cfg in "1000000 512 384" "1000000 1920 1080"; do
      if [ $lib = default ]; then unset ARTDECO_HIP_LIB; else export ARTDECO_HIP_LIB=$PWD/artdeco_amd/lib/libartdeco_hip.$lib.so; fi
      timeout 300 python tools/lab/stage_times.py $cfg lod_params_bwd 2>&1 | tail -1 | sed "s/^/$lib /" >> gpurun_out/r04_ab_lod_late2.txt
    done
  done
done
cat gpurun_out/r04_ab_lod_late2.txt

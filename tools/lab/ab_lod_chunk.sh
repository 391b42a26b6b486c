#!/bin/bash
# lod_params_bwd on 64- vs 32-Gaussian chunks (ADK_LOD_BWD_CHUNK, read per launch), same box, alternating
set -u
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_fused_glue.py -q -m gpu -k "lod_backward_forms" 2>&1 | tail -3 > gpurun_out/r04_ab_lod_chunk.txt
for rep in 1 2; do
  for c in 64 32; do
    for cfg in "1000000 512 384" "1000000 1920 1080"; do
      ADK_LOD_BWD_CHUNK=$c timeout 300 python tools/lab/stage_times.py $cfg lod_params_bwd 2>&1 | tail -1 | sed "s/^/chunk=$c /" >> gpurun_out/r04_ab_lod_chunk.txt
    done
  done
done
cat gpurun_out/r04_ab_lod_chunk.txt

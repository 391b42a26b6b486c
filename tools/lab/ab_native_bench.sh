#!/bin/bash
# A/B of the frame loop with / without the one-call optimisation step (same box, alternating)
set -u
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_fused_glue.py tests/test_reference_scene_model.py tests/test_psnr_proxy.py tests/test_multigpu.py tests/test_bench_contract.py -q -m gpu -x > gpurun_out/r04_native_suites.log 2>&1
tail -4 gpurun_out/r04_native_suites.log
: > gpurun_out/r04_ab_native_bench.txt
for rep in 1 2; do
  for v in 1 0; do
    ARTDECO_AMD_NATIVE_STEP=$v timeout 600 python bench.py --no-extra-configs --no-cpu-baseline --no-frontend 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('native=$v', 'frames/s', round(d['value'],2), 'ms/opt-step', round(d['ms_per_optimisation_step_incl_frame_overheads'],4), 'raster_bwd', d['stage_ms'].get('raster_bwd'), 'stages', len(d['stage_ms']))" >> gpurun_out/r04_ab_native_bench.txt
  done
done
cat gpurun_out/r04_ab_native_bench.txt

#!/bin/bash
# frame loop with capacity-rounded step plans (default) vs one plan per exact N (same box, alternating)
set -u
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_native_step.py -q -m gpu 2>&1 | tail -2 > gpurun_out/r04_ab_plan_grain.txt
for rep in 1 2 3; do
  for g in 65536 1; do
    ARTDECO_AMD_PLAN_GRAIN=$g timeout 600 python bench.py --no-extra-configs --no-cpu-baseline --no-frontend 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('grain=$g', 'frames/s', round(d['value'],2), 'ms/opt-step', round(d['ms_per_optimisation_step_incl_frame_overheads'],4), 'frame stages', {k: round(v['ms_per_frame'],3) for k,v in d['frame_stage_ms'].items()})" >> gpurun_out/r04_ab_plan_grain.txt
  done
done
cat gpurun_out/r04_ab_plan_grain.txt

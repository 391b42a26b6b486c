#!/bin/bash
# DEV TOOL (round 5): the two-halves raster backward as two single-wave workgroups per tile (default) against ONE 128-thread workgroup per
# tile sharing the staging, the parked table and the flush (ADK_RASTER_BWD_WG2=1): parity, time (alternating, same box) and HBM counters.
#   bash tools/lab/ab_bwd_wg2.sh > gpurun_out/r05_ab_bwd_wg2.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
echo "# parity of the WG2 form (fp64-autograd oracle at the BASELINE sizes + the wave forms against each other)"
ADK_RASTER_BWD_WG2=1 timeout 900 python -m pytest tests/test_raster.py -x -q -m gpu -k "backward or baseline_sizes or waves_per_tile or northstar_sizes_in_every_wave_form" 2>&1 | tail -1
for rep in 1 2 3; do
  for v in 0 1; do
    echo -n "WG2=$v "; ADK_RASTER_BWD_WG2=$v timeout 300 python tools/lab/stage_times.py 1000000 1920 1080 raster_bwd 2>&1 | tail -1
  done
done
for v in 0 1; do
  echo -n "WG2=$v 4M/2592x1944 "; ADK_RASTER_BWD_WG2=$v timeout 300 python tools/lab/stage_times.py 4000000 2592 1944 raster_bwd 2>&1 | tail -1
done
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU"; do
    rm -rf /tmp/pv
    ADK_RASTER_BWD_WG2=$v timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pv -o b -- python $ROOT/tools/lab/stage_times.py 1000000 1920 1080 raster_bwd > /tmp/pv.log 2>&1 || tail -3 /tmp/pv.log
    echo -n "WG2=$v counters: "; python $ROOT/tools/lab/pmc_one.py /tmp/pv raster_bwd_kernel 1000
  done
done

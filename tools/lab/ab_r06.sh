#!/bin/bash
# DEV TOOL (round 6): same-box A/B of library variants (tools/lab/ab_step.py --build name=flags ...) on the raster stages.
#   bash tools/lab/ab_r06.sh "default old ..." [check]   > gpurun_out/r06_ab_<what>.txt
cd "$(dirname "$0")/../.."
LIBDIR=artdeco_amd/lib
VARIANTS="${1:-default old}"
lib_of() { if [ "$1" = default ]; then echo $LIBDIR/libartdeco_hip.so; else echo $LIBDIR/libartdeco_hip.$1.so; fi; }
if [ "$2" = check ]; then
  for v in $VARIANTS; do
    ARTDECO_HIP_LIB=$(lib_of $v) timeout 1200 python -m pytest tests/test_raster.py -x -q -m gpu 2>&1 | tail -1 | sed "s/^/[check $v] /"
  done
fi
for rep in 1 2; do
  for v in $VARIANTS; do
    ARTDECO_HIP_LIB=$(lib_of $v) timeout 300 python tools/lab/stage_times.py 1000000 1920 1080 raster_bwd,raster_fwd 2>&1 | tail -1
  done
  for v in $VARIANTS; do
    ARTDECO_HIP_LIB=$(lib_of $v) timeout 300 python tools/lab/stage_times.py 1000000 512 384 raster_bwd,raster_fwd 2>&1 | tail -1
  done
done

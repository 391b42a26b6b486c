#!/bin/bash
# DEV TOOL (round 4): same-box A/B of raster_bwd's first-touch / paired-reduction forms (library variants built by
# `tools/lab/ab_step.py --build first=... pair=... firstpair=... firstpair5=...`), each in the whole-tile and the two-halves form at
# 1 M / 1080p and in the default (quadrant) form at 1 M / 512x384.  Run on the GPU box:  bash tools/lab/ab_bwd_r04.sh > gpurun_out/r04_ab_bwd.txt
cd "$(dirname "$0")/../.."
LIBDIR=artdeco_amd/lib
VARIANTS="${VARIANTS:-default first pair firstpair firstpair5}"
lib_of() { if [ "$1" = default ]; then echo $LIBDIR/libartdeco_hip.so; else echo $LIBDIR/libartdeco_hip.$1.so; fi; }
echo "# parity of every variant (fp64-autograd oracle on small frames + the three wave forms against each other)"
for v in $VARIANTS; do
  [ "$v" = default ] && continue
  ARTDECO_HIP_LIB=$(lib_of $v) timeout 600 python -m pytest tests/test_raster.py -x -q -m gpu -k "test_backward_matches_fp64_autograd_oracle or test_waves_per_tile_forms_composite_the_same_pixels" 2>&1 | tail -1 | sed "s/^/[check $v] /"
done
for rep in 1 2; do
  for form in 0 2; do
    for v in $VARIANTS; do
      echo -n "1080p split_bwd=$form "
      ARTDECO_HIP_LIB=$(lib_of $v) ADK_RASTER_SPLIT_BWD=$form timeout 300 python tools/lab/stage_times.py 1000000 1920 1080 raster_bwd,raster_fwd 2>&1 | tail -1
    done
  done
  for v in $VARIANTS; do
    echo -n "512x384 default-split "
    ARTDECO_HIP_LIB=$(lib_of $v) timeout 300 python tools/lab/stage_times.py 1000000 512 384 raster_bwd,raster_fwd 2>&1 | tail -1
  done
done

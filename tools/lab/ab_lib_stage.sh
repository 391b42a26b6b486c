#!/bin/bash
# same-box A/B of one stage between the in-tree library and a variant build:  bash tools/lab/ab_lib_stage.sh <variant> <stage> <out-tag> [pytest -k expr]
set -u
V=$1; S=$2; TAG=$3; K=${4:-}
mkdir -p gpurun_out
OUT=gpurun_out/r04_ab_${TAG}.txt
: > $OUT
if [ -n "$K" ]; then timeout 900 python -m pytest tests -q -m gpu -k "$K" 2>&1 | tail -2 >> $OUT; fi
for rep in 1 2; do
  for lib in default $V; do
    for cfg in "1000000 512 384" "1000000 1920 1080"; do
      if [ $lib = default ]; then unset ARTDECO_HIP_LIB; else export ARTDECO_HIP_LIB=$PWD/artdeco_amd/lib/libartdeco_hip.$lib.so; fi
      timeout 300 python tools/lab/stage_times.py $cfg $S 2>&1 | tail -1 | sed "s/^/$lib /" >> $OUT
    done
  done
done
cat $OUT

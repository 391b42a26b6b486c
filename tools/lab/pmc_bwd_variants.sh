#!/bin/bash
# DEV TOOL (round 4): SQ counters of raster_bwd_kernel per library variant and wave form, on the stationary 1 M / 1080p step.
#   bash tools/lab/pmc_bwd_variants.sh "default:0 firstpair:0 firstpair:2" > gpurun_out/r04_pmc_bwd_variants.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
A="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"
B="SQ_INST_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_BRANCH"
C="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_IFETCH SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_ADD_F32"
for spec in ${1:-default:0}; do
  v=${spec%%:*}; form=${spec##*:}
  if [ "$v" = default ]; then lib=$ROOT/artdeco_amd/lib/libartdeco_hip.so; else lib=$ROOT/artdeco_amd/lib/libartdeco_hip.$v.so; fi
  for pass in A B C; do
    rm -rf /tmp/pv
    ARTDECO_HIP_LIB=$lib ADK_RASTER_SPLIT_BWD=$form timeout 600 rocprofv3 --pmc ${!pass} --kernel-trace --output-format csv -d /tmp/pv -o b -- \
      python $ROOT/tools/lab/stage_times.py 1000000 1920 1080 raster_bwd > /tmp/pv.log 2>&1 || tail -3 /tmp/pv.log
    echo -n "$v split_bwd=$form pass $pass: "; python $ROOT/tools/lab/pmc_one.py /tmp/pv raster_bwd_kernel 1000
  done
done

#!/bin/bash
# DEV TOOL (round 4): SQ counters of one kernel on the stationary 1 M / 1080p step, three --pmc passes.
#   bash tools/lab/pmc_kernel.sh lod_params_bwd_kernel [min_grid] > gpurun_out/r04_pmc_<kernel>.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
K=${1:-lod_params_bwd_kernel}; MING=${2:-1000}
cd /tmp && export TMPDIR=/tmp
A="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES"
B="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
C="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"
for pass in A B C; do
  rm -rf /tmp/pv
  timeout 600 rocprofv3 --pmc ${!pass} --kernel-trace --output-format csv -d /tmp/pv -o b -- python $ROOT/tools/lab/stage_times.py 1000000 1920 1080 lod_params_bwd > /tmp/pv.log 2>&1 || tail -3 /tmp/pv.log
  echo -n "pass $pass: "; python $ROOT/tools/lab/pmc_one.py /tmp/pv $K $MING
done

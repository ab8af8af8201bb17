#!/bin/bash
# PMC passes (separate runs, kernel-trace only) for s3::step_kernel on tools/s3_run.py: run on the GPU box from the repo root.
#   tools/pmc_s3.sh 2|3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_WAVES SQ_INSTS_SALU" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_s3c$1/p$i -- python3 $R/tools/s3_run.py $1 ${2:-4} > $R/gpurun_out/pmc_s3c$1_$i.log 2>&1 || { echo "pass $i failed"; tail -3 $R/gpurun_out/pmc_s3c$1_$i.log; exit 1; }
done

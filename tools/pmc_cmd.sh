#!/bin/bash
# PMC passes (separate runs, kernel-trace only) for an arbitrary python target: run on the GPU box from the repo root.
#   tools/pmc_cmd.sh TAG tools/cfg5_run.py f16x3 128        -> gpurun_out/pmc_TAG/p*/...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=$1; shift
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "SQ_WAVES SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_IFETCH" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$TAG/p$i -- python3 $R/"$@" > $R/gpurun_out/pmc_${TAG}_$i.log 2>&1 || { echo "pass $i failed"; tail -3 $R/gpurun_out/pmc_${TAG}_$i.log; exit 1; }
done

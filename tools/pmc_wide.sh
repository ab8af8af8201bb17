#!/bin/bash
# PMC passes over the d=256 bf16 rollout (wide fused step kernel): run on the GPU box from the repo root.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
           "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_wide/p$i -- python3 $R/bench.py --d-model 256 --d-ff 1024 --heads 8 --precision bf16 --steps 1 --warmup 1 --graph 0 --no-cpu-baseline --train-steps 0 > $R/gpurun_out/pmc_wide_$i.log 2>&1 || exit 1
done

#!/bin/bash
# PMC passes (separate runs, kernel-trace only) for the two dominant kernels: run on the GPU box from the repo root.
#   tools/pmc_all.sh d32   -> fused::rollout_f32_kernel   (default bench config)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
#   tools/pmc_all.sh x3    -> x3::layer_kernel            (--d-model 256 --d-ff 1024 --heads 8 --precision f16x3)
if [ "$1" = "d256" ]; then ARGS="--d-model 256 --d-ff 1024 --heads 8 --precision bf16";
elif [ "$1" = "x3" ]; then ARGS="--d-model 256 --d-ff 1024 --heads 8 --precision f16x3";
#   tools/pmc_all.sh s3    -> s3::step_kernel             (--precision f16x3: the headline shape at reference precision on the f16 pipe)
elif [ "$1" = "s3" ]; then ARGS="--precision f16x3"; else ARGS=""; fi
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$1$TAG/p$i -- python3 $R/bench.py $ARGS --steps 1 --warmup 1 --graph 0 --no-cpu-baseline --train-steps 0 > $R/gpurun_out/pmc_$1${TAG}_$i.log 2>&1 || exit 1
done

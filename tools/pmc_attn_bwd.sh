#!/bin/bash
# Counters of the attention backward at the cfg3 shape (one pass per counter set, --pmc only with --kernel-trace); GPU box, repo root.
# usage: bash tools/pmc_attn_bwd.sh <out-subdir> [ALINE_DBG value]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
[ -n "$2" ] && export ALINE_DBG=$2
mkdir -p $O
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -- python3 $R/tools/train_cfg3.py 128 > $O/log_$i.txt 2>&1 || exit 1
done
python3 $R/tools/pmc_summary.py $O attention_bwd > $O/summary.txt 2>&1
cat $O/summary.txt

#!/bin/bash
# same-box A/B of library variants on the d = 256 training step: kernel stats of tools/d256_train_run.py per variant
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for name in ${VARIANTS:-base head occ1}; do
  lib=$R/aline_amd/csrc/variants/lib_$name.so; [ "$name" = base ] && lib=$R/aline_amd/csrc/libaline_hip.so
  echo "== $name"; rm -rf $R/gpurun_out/dw_$name
  ALINE_HIP_LIB=$lib timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/dw_$name -- python3 $R/tools/d256_train_run.py ${DW_ARGS} > $R/gpurun_out/dw_$name.log 2>&1 || { echo FAILED; tail -3 $R/gpurun_out/dw_$name.log; continue; }
  python3 $R/tools/prof_stats.py $R/gpurun_out/dw_$name 3
done

#!/bin/bash
# same-box A/B of library variants on the training steps: kernel stats of the headline step and of the cfg3 step (GPU box, repo root)
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for name in ${VARIANTS:-base head}; do
  lib=$R/aline_amd/csrc/variants/lib_$name.so; [ "$name" = base ] && lib=$R/aline_amd/csrc/libaline_hip.so
  for tgt in train_headline_run train_cfg3; do
    echo "== $name $tgt"; rm -rf $R/gpurun_out/tab_${name}_$tgt
    ALINE_HIP_LIB=$lib timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/tab_${name}_$tgt -- python3 $R/tools/$tgt.py > $R/gpurun_out/tab_${name}_$tgt.log 2>&1 || { echo FAILED; tail -3 $R/gpurun_out/tab_${name}_$tgt.log; continue; }
    grep "train step" $R/gpurun_out/tab_${name}_$tgt.log
    python3 $R/tools/prof_stats.py $R/gpurun_out/tab_${name}_$tgt 12 | grep "acqb::bwd\|gmmb::bwd"
  done
done

#!/bin/bash
# same-box A/B of the interleaved chunk schedule (x3_impl.h chunk_pipe): default build (x5 interleaved) vs variants
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1; mkdir -p $O
for name in ${VARIANTS:-base noilv v1 v2}; do
  lib=$R/aline_amd/csrc/variants/lib_$name.so; [ "$name" = base ] && lib=$R/aline_amd/csrc/libaline_hip.so
  echo "== $name"
  ALINE_HIP_LIB=$lib timeout -k 10 200 python3 $R/tools/config_bench.py --configs 5 --precs f16x3 2>&1 | grep -v amdgpu.ids | tee -a $O/$name.txt | python3 -c "
import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    print({k: r[k] for k in r if k in ('config','F','d_ff','ms_per_rollout','path','frac','precision')})"
  ALINE_HIP_LIB=$lib timeout -k 10 200 python3 $R/tools/x3_time.py 2>&1 | grep -v amdgpu.ids | tee -a $O/$name.txt | tail -2
done

#!/bin/bash
# session 16: split K / V kernel: parity + cfg5 timing
cd $GRAFT_REPO_ROOT
O=gpurun_out/s16; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_x5_gpu.py tests/test_x3_gpu.py -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/tests.log
[ $rc -ne 0 ] && exit 0
run() { python tools/config_bench.py --configs 5 --precs f16x3 2>$O/err.log | python -c 'import json,sys
for l in sys.stdin:
    d=json.loads(l); print(round(d["ms_per_rollout"],3), d["path"], end=" | ")'; }
for v in base prev base prev; do
  lib=$PWD/aline_amd/csrc/variants/lib_$v.so; [ $v = base ] && lib=$PWD/aline_amd/csrc/libaline_hip.so
  echo "cfg5 $v: $(ALINE_HIP_LIB=$lib run)" | tee -a $O/kv_split.txt
done

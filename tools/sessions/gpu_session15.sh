#!/bin/bash
# session 15: XCD-aware tile order of the layer kernels: parity + A/B (x5 at cfg5, x3 at the d = 256 headline shape)
cd $GRAFT_REPO_ROOT
O=gpurun_out/s15; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_x5_gpu.py tests/test_x3_gpu.py -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/tests.log
[ $rc -ne 0 ] && exit 0
run() { python tools/config_bench.py --configs 5 --precs f16x3 2>$O/err.log | python -c 'import json,sys
for l in sys.stdin:
    d=json.loads(l); print(round(d["ms_per_rollout"],3), d["path"], end=" | ")'; }
for v in base prev base prev; do
  lib=$PWD/aline_amd/csrc/variants/lib_$v.so; [ $v = base ] && lib=$PWD/aline_amd/csrc/libaline_hip.so
  echo "cfg5 $v: $(ALINE_HIP_LIB=$lib run)" | tee -a $O/xcd_order.txt
  echo "d256 $v: $(ALINE_HIP_LIB=$lib python tools/x3_time.py 2>>$O/err.log)" | tee -a $O/xcd_order.txt
done

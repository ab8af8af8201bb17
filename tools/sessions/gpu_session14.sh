#!/bin/bash
# session 14: x5 K / V prefetch depth A/B at cfg5 (F = 128)
cd $GRAFT_REPO_ROOT
O=gpurun_out/s14; mkdir -p $O
run() { python tools/config_bench.py --configs 5 --precs f16x3 2>$O/err.log | head -1 | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(round(d["ms_per_rollout"],3), d["path"])'; }
for v in base kv2 kv3 prev base kv2 kv3 prev; do
  lib=$PWD/aline_amd/csrc/variants/lib_$v.so; [ $v = base ] && lib=$PWD/aline_amd/csrc/libaline_hip.so
  echo "$v: $(ALINE_HIP_LIB=$lib run)" | tee -a $O/kv_ahead.txt
done
timeout -k 10 300 python -m pytest tests/test_x5_gpu.py -m gpu -x -q 2>&1 | tail -2

#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/s2; mkdir -p $O
timeout -k 10 120 python tools/ws_debug.py > $O/ws_debug.log 2>&1; echo "ws_debug rc=$?"
tail -4 $O/ws_debug.log | cut -c1-400
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "pytest rc=$?"
tail -15 $O/gpu_tests.log
for v in base x3new x3_nonext x3_noresid x3_nokvpf base x3new; do
  lib=$PWD/aline_amd/csrc/variants/lib_$v.so; [ $v = base ] && lib=$PWD/aline_amd/csrc/libaline_hip.so
  ALINE_HIP_LIB=$lib timeout -k 10 200 python tools/x3_time.py 2>> $O/x3_time.err | tee -a $O/x3_time.jsonl
done
ALINE_HIP_LIB=$PWD/aline_amd/csrc/variants/lib_x3new.so timeout -k 10 600 python -m pytest tests/test_x3_gpu.py tests/test_r2_gpu.py tests/test_hip_parity.py -m gpu -q -k "x3 or deep or d256" > $O/x3new_tests.log 2>&1; echo "x3new tests rc=$?"
tail -5 $O/x3new_tests.log

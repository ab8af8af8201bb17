#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/s4; mkdir -p $O
for v in base old4 stag both base; do
  lib=$PWD/aline_amd/csrc/variants/lib_$v.so; [ $v = base ] && lib=$PWD/aline_amd/csrc/libaline_hip.so
  ALINE_HIP_LIB=$lib timeout -k 10 120 python tools/x3_time.py 2>> $O/x3_time.err | tee -a $O/x3_time.jsonl || { echo "$v FAILED/timeout"; exit 1; }
done
for v in stag both; do
ALINE_HIP_LIB=$PWD/aline_amd/csrc/variants/lib_$v.so timeout -k 10 300 python -m pytest tests/test_x3_gpu.py tests/test_r2_gpu.py tests/test_hip_parity.py -m gpu -q -k "x3 or deep or d256" > $O/tests_$v.log 2>&1; echo "$v tests rc=$?"; tail -3 $O/tests_$v.log
done

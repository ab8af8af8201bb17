#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/s5; mkdir -p $O
for v in base bufdma base bufdma; do
  lib=$PWD/aline_amd/csrc/variants/lib_$v.so; [ $v = base ] && lib=$PWD/aline_amd/csrc/libaline_hip.so
  ALINE_HIP_LIB=$lib timeout -k 10 120 python tools/x3_time.py 2>> $O/x3_time.err | tee -a $O/x3_time.jsonl || { echo "$v FAILED/timeout"; exit 1; }
done
timeout -k 10 900 python -m pytest tests/test_x3_gpu.py tests/test_r2_gpu.py tests/test_hip_parity.py tests/test_backward_gpu.py tests/test_driver.py tests/test_range_guard_gpu.py -m gpu -q --durations=12 > $O/tests.log 2>&1; echo "tests rc=$?"; tail -25 $O/tests.log
ALINE_HIP_LIB=$PWD/aline_amd/csrc/variants/lib_bufdma.so timeout -k 10 300 python -m pytest tests/test_x3_gpu.py -m gpu -q > $O/tests_bufdma.log 2>&1; echo "bufdma tests rc=$?"; tail -3 $O/tests_bufdma.log

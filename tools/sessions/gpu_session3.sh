#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/s3; mkdir -p $O
timeout -k 10 120 python tools/ws_debug.py > $O/ws_debug.log 2>&1; echo "ws_debug rc=$?"; tail -2 $O/ws_debug.log | cut -c1-200
for v in base ks0 kv2 prio nobar nodma base; do
  lib=$PWD/aline_amd/csrc/variants/lib_$v.so; [ $v = base ] && lib=$PWD/aline_amd/csrc/libaline_hip.so
  ALINE_HIP_LIB=$lib timeout -k 10 200 python tools/x3_time.py 2>> $O/x3_time.err | tee -a $O/x3_time.jsonl
done
timeout -k 10 600 python -m pytest tests/test_x3_gpu.py tests/test_r2_gpu.py tests/test_hip_parity.py tests/test_range_guard_gpu.py tests/test_workspace_gpu.py -m gpu -q -k "x3 or deep or d256 or range or workspace" > $O/x3_tests.log 2>&1; echo "x3 tests rc=$?"; tail -5 $O/x3_tests.log
ALINE_HIP_LIB=$PWD/aline_amd/csrc/variants/lib_stamps.so timeout -k 10 200 python tools/x3_stamps.py 30 > $O/x3_stamps.txt 2>&1; echo "stamps rc=$?"; tail -9 $O/x3_stamps.txt
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/tools/x3_time.py 1000 30 2 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1); echo "prof rc=$?"
python3 tools/prof_stats.py $O/prof 8

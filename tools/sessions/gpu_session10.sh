#!/bin/bash
# session 10: cfg3 on the 12-wave head-by-head variant (A/B) + its parity tests
cd $GRAFT_REPO_ROOT
O=gpurun_out/s10; mkdir -p $O
run() { python tools/config_bench.py --configs $1 --precs f16x3 2>$O/err.log | grep '"d": 32' | head -1 | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(round(d["ms_per_rollout"],3), d["path"])'; }
echo "cfg3 default: $(run 3)" | tee $O/cfg3_ab.txt
echo "cfg3 12 waves lean: $(ALINE_DBG=S3_WAVES=12 run 3)" | tee -a $O/cfg3_ab.txt
echo "cfg3 default: $(run 3)" | tee -a $O/cfg3_ab.txt
echo "cfg3 12 waves lean: $(ALINE_DBG=S3_WAVES=12 run 3)" | tee -a $O/cfg3_ab.txt
ALINE_DBG=S3_WAVES=12 timeout -k 10 600 python -m pytest tests/test_s3_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q > $O/s3_lean_tests.log 2>&1; echo "lean tests rc=$?"; tail -3 $O/s3_lean_tests.log

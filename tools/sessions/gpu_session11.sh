#!/bin/bash
# session 11: x5 path (d = 512) first light: parity tests, the cfg5 fixtures, timing
cd $GRAFT_REPO_ROOT
O=gpurun_out/s11; mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_x5_gpu.py -m gpu -x -q > $O/x5_tests.log 2>&1; rc=$?; echo "x5 tests rc=$rc"; tail -15 $O/x5_tests.log
[ $rc -ne 0 ] && exit 0
timeout -k 10 500 python -m pytest tests/test_x3_gpu.py tests/test_r2_gpu.py tests/test_hip_parity.py -m gpu -x -q > $O/x3_tests.log 2>&1; echo "x3/r2/parity rc=$?"; tail -5 $O/x3_tests.log
timeout -k 10 300 python tools/config_bench.py --configs 5 --precs f16x3 > $O/cfg5.jsonl 2> $O/cfg5.err; echo "cfg5 rc=$?"; python - <<'PY'
import json
for l in open("gpurun_out/s11/cfg5.jsonl"):
    d = json.loads(l); print(d["config"], round(d["ms_per_rollout"], 2), d["path"], round(d["roofline"]["frac"], 4))
PY
ALINE_DBG=DISABLE_X3 timeout -k 10 300 python tools/config_bench.py --configs 5 --precs f16x3 > $O/cfg5_generic.jsonl 2>> $O/cfg5.err; python - <<'PY'
import json
for l in open("gpurun_out/s11/cfg5_generic.jsonl"):
    d = json.loads(l); print(d["config"], round(d["ms_per_rollout"], 2), d["path"], round(d["roofline"]["frac"], 4))
PY

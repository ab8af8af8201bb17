#!/bin/bash
# session 23: saved activations for the backward: parity + train step + headline unchanged
cd $GRAFT_REPO_ROOT
O=gpurun_out/s23; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_backward_gpu.py -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "backward tests rc=$rc"; tail -8 $O/tests.log
[ $rc -ne 0 ] && exit 0
timeout -k 10 600 python bench.py --no-cpu-baseline --no-f32 --no-query-gmm --no-d512 --no-d256 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/s23/bench.json"))
print("value", d["value"], "ms", d["ms_per_step"], "kernel", d["roofline"]["kernel"], d["roofline"]["kernel_ms_per_launch"], "train", d["train_step"]["ms_per_step"])
PY
ALINE_DBG=NO_BWD_SAVED_ACTS timeout -k 10 600 python bench.py --no-cpu-baseline --no-f32 --no-query-gmm --no-d512 --no-d256 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print("recompute: train", d["train_step"]["ms_per_step"])'

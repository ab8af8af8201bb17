#!/bin/bash
# session 20: wide GMM backward kernel: parity + d256 train step
cd $GRAFT_REPO_ROOT
O=gpurun_out/s20; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_backward_gpu.py -m gpu -x -q -k "wide_gmm or gradients_match" > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $O/tests.log
[ $rc -ne 0 ] && exit 0
timeout -k 10 600 python bench.py --no-cpu-baseline --no-f32 --no-query-gmm --no-d512 --steps 2 --warmup 1 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open("gpurun_out/s20/bench.json"))
print("d256 train_step", d["d256"]["f16x3"].get("train_step"))
PY

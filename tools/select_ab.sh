#!/bin/bash
# same-box A/B of the design-selection kernels on the headline rollout (GPU box, repo root)
A="--steps 30 --warmup 3 --no-cpu-baseline --no-d256 --no-d512 --no-f32 --no-query-gmm --train-steps 0"
for i in 1 2; do
  echo "wave:      $(python3 bench.py $A 2>/dev/null | python3 -c 'import sys,json; r=json.loads(sys.stdin.read()); print(round(r["ms_per_step"],4), round(r["sustained_ms_per_step"],4), round(r["value"]/1e9,4))')"
  echo "workgroup: $(ALINE_DBG=SELECT_WORKGROUP python3 bench.py $A 2>/dev/null | python3 -c 'import sys,json; r=json.loads(sys.stdin.read()); print(round(r["ms_per_step"],4), round(r["sustained_ms_per_step"],4), round(r["value"]/1e9,4))')"
done

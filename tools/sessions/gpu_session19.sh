#!/bin/bash
# session 19: full GPU suite + smoke on the x5 build, bench line, per-config table, cfg5 kernel stats
cd $GRAFT_REPO_ROOT
O=gpurun_out/s19; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -q -x --durations=8 > $O/gpu_tests.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -14 $O/gpu_tests.log
timeout -k 10 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
[ $rc -ne 0 ] && exit 0
timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; python - <<'PY'
import json
d = json.load(open("gpurun_out/s19/bench.json"))
print("value", d["value"], "ms", d["ms_per_step"], "sustained", d.get("sustained_ms_per_step"), "train", d["train_step"]["ms_per_step"])
for k in ("d256", "d512"):
    if k in d:
        x = d[k]["f16x3"]; print(k, round(x["ms_per_rollout"], 2), x["roofline"]["kernel"], round(x["roofline"]["frac"], 4), round(x["whole_rollout"]["frac"], 4))
PY
timeout -k 10 600 python tools/config_bench.py > $O/config_bench.jsonl 2> $O/config_bench.err; echo "config bench rc=$?"
python - <<'PY'
import json
for l in open("gpurun_out/s19/config_bench.jsonl"):
    d = json.loads(l); print(d["config"][:70], d["precision"], round(d["ms_per_rollout"], 2), d["path"])
PY
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_cfg5 -- python3 $GRAFT_REPO_ROOT/tools/cfg5_run.py f16x3 128 > $GRAFT_REPO_ROOT/$O/cfg5_prof.log 2>&1); echo "cfg5 prof rc=$?"
python3 tools/prof_stats.py $O/prof_cfg5 8

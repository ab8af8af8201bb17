#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/s22; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_x5_gpu.py tests/test_x3_gpu.py -m gpu -x -q > $O/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/tests.log
python tools/x3_time.py 2>/dev/null
python tools/config_bench.py --configs 5 --precs f16x3 2>/dev/null | python -c 'import json,sys
for l in sys.stdin:
    d=json.loads(l); print(round(d["ms_per_rollout"],3), d["path"])'
cat > /tmp/train_run.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from aline_amd import Aline, Embedder, Encoder, OutputHead
from aline_amd.tasks import HiddenLocation
from aline_amd.train import train_step
torch.manual_seed(0)
dev = torch.device("cuda")
m = Aline(Embedder(2, 1, 32, 128, 2, "theta"), Encoder(32, 128, 4, 0.0, 3), OutputHead(2, 1, 32, 128)).cuda().set_precision("f16x3").train()
batch = HiddenLocation(n_query_init=200, device=dev).sample_batch(1000)
opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
for _ in range(4):
    train_step(m, batch, 30, optimizer=opt)
    torch.cuda.synchronize()
PY
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_train -- python3 /tmp/train_run.py > $GRAFT_REPO_ROOT/$O/train_run.log 2>&1); echo "rc=$?"
python3 tools/prof_stats.py $O/prof_train 20

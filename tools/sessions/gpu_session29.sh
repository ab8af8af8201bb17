#!/bin/bash
# session 29: dW products with 64-column blocks: parity + wide-model training steps
cd $GRAFT_REPO_ROOT
timeout -k 10 700 python -m pytest tests/test_backward_gpu.py -m gpu -x -q 2>&1 | tail -2
timeout -k 10 300 python tools/train_cfg5.py 2>&1 | tail -1
ALINE_DBG=BWD_DW_TK2 timeout -k 10 300 python tools/train_cfg5.py 2>&1 | tail -1
cat > /tmp/t256.py <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from aline_amd import Aline, Embedder, Encoder, OutputHead
from aline_amd.tasks import HiddenLocation
from aline_amd.train import train_step
torch.manual_seed(0)
dev = torch.device("cuda")
m = Aline(Embedder(2, 1, 256, 1024, 2, "theta"), Encoder(256, 1024, 8, 0.0, 3), OutputHead(2, 1, 256, 1024)).cuda().set_precision("f16x3").train()
batch = HiddenLocation(n_query_init=200, device=dev).sample_batch(1000)
opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
train_step(m, batch, 30, optimizer=opt); torch.cuda.synchronize()
t0 = time.time(); terms, ro = train_step(m, batch, 30, optimizer=opt); torch.cuda.synchronize(); print("d256 train step ms", (time.time() - t0) * 1e3, float(terms["loss"]))
PY
python /tmp/t256.py 2>&1 | tail -1
ALINE_DBG=BWD_DW_TK2 python /tmp/t256.py 2>&1 | tail -1
timeout -k 10 300 python tools/train_cfg3.py 2>&1 | tail -1

"""Latency of the drop-in per-step API: `Aline.forward(batch)` + `Task.update_batch` as the reference's own loop calls
them (train_aline.py:84-88, utils/eval.py:28-30), eval mode, location_finding.   python tools/step_latency.py"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from aline_amd import Aline, Embedder, Encoder, OutputHead  # noqa: E402
from aline_amd.tasks import HiddenLocation  # noqa: E402

dev = torch.device("cuda")
torch.manual_seed(0)
model = Aline(Embedder(2, 1, 32, 128, 2, "theta"), Encoder(32, 128, 4, 0.0, 3), OutputHead(2, 1, 32, 128)).to(dev).eval()
for B in (20, 200, 1000):
    task = HiddenLocation(n_query_init=200, device=dev)
    T = 30
    with torch.no_grad():
        for rep in range(3):
            batch = task.sample_batch(B)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for t in range(T):
                out = model.forward(batch)
                batch = task.update_batch(batch, out.design_out.idx)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
    print(json.dumps({"B": B, "T": T, "ms_per_forward_plus_update": dt / T * 1e3, "designs_per_s": B * T * 200 / dt}))

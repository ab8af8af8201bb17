"""Sanity run of the training driver (aline_amd.driver.train) on a small location-finding problem: the prediction loss
must fall during burn-in and keep falling once the design loss is switched on.   python tools/train_convergence.py [f32|f16x3] [d F H]
(d = 32 / F = 128 / H = 4: fused backward kernels; any other width: the per-op pipeline)"""
import json
import os
import random
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from aline_amd import Aline, Embedder, Encoder, OutputHead  # noqa: E402
from aline_amd.driver import train  # noqa: E402
from aline_amd.tasks import HiddenLocation  # noqa: E402


class Cfg(dict):
    __getattr__ = dict.get


torch.manual_seed(0); random.seed(0)
dev = torch.device("cuda")
d, F, H = (int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (32, 128, 4)
model = Aline(Embedder(2, 1, d, F, 2, "theta"), Encoder(d, F, H, 0.0, 3), OutputHead(2, 1, d, F)).to(dev)
model.set_precision(sys.argv[1] if len(sys.argv) > 1 else "f32")       # "f16x3": rollouts on the s3 path, backward in exact fp32
task = HiddenLocation(n_query_init=50, device=dev)
cfg = Cfg(optimizer="AdamW", lr=1e-3, max_epoch=400, burning_epoch=200, checkpoint=0, output_dir="/tmp/aline_probe",
          file_name="probe.pth", T=10, min_T=10, alpha=1.0, gamma=1.0, clip_grads=True, batch_size=256, verbose=10 ** 9,
          task=Cfg(mask_type=["all"], embedding_type="theta", n_target_data=0, n_target_theta=2, n_query_init=50))
recs = train(cfg, model, task)
for lo in range(0, 400, 50):
    w = recs[lo:lo + 50]
    print(json.dumps({"epochs": f"{lo}-{lo + 49}", "predict_loss": sum(r["predict_loss"] for r in w) / len(w),
                      "design_loss": sum(r["design_loss"] for r in w) / len(w),
                      "ms_per_epoch": 1e3 * sum(r["seconds"] for r in w) / len(w)}))

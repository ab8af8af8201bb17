#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of the fused rollout kernel (s_memtime stamps, workgroup 0).
Run on the GPU box:  python tools/stamps.py [--batch 1000]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aline_amd import Aline, Embedder, Encoder, OutputHead, _lib  # noqa: E402
from aline_amd.rollout import Rollout  # noqa: E402
from aline_amd.tasks import HiddenLocation  # noqa: E402
_lib.lib.aline_debug_set_flags(_lib.DBG["FUSED_STAMPS"])

PH = ["key list", "x0 load", "weight stream+bar", "pre-pass+bar", "main pass", "wait slowest",
      "head stream", "acq MLP", "bar after acq", "select / GMM", "bar after sel", "GMM epilogue+bar"]

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1000)
ap.add_argument("--T", type=int, default=30)
args = ap.parse_args()
dev = torch.device("cuda")
model = Aline(Embedder(2, 1, 32, 128, 2, "theta"), Encoder(32, 128, 4, 0.0, 3),
              OutputHead(2, 1, 32, 128)).to(dev)
model.train()
task = HiddenLocation(device=dev)
ro = Rollout(model, task.sample_batch(args.batch), args.T, select="sample")
off = _lib.lib.aline_debug_stamps_offset(C.byref(ro.m), C.byref(ro.r))
ro.run(); torch.cuda.synchronize()
ro.ws[off:off + 8 * 16 * 8].zero_()
ro.run(); torch.cuda.synchronize()
st = ro.ws[off:off + 8 * 16 * 8].view(torch.int64).reshape(8, 16).cpu()
tot = st[:4].sum(1).float()
print(f"cycles per rollout (wave 0..3 of workgroup 0): {tot.tolist()}")
for i, nm in enumerate(PH):
    row = st[:4, i].float()
    print(f"{nm:20s} " + "  ".join(f"{v/1e3:9.1f}k ({100*v/t:4.1f}%)" for v, t in zip(row.tolist(), tot.tolist())))

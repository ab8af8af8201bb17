#!/usr/bin/env python3
"""Diagnostic: per-phase s_memtime shares of the wide fused step kernel (workgroup 0, all steps summed).
Run on the GPU box:  python tools/wide_stamps.py [--batch 1000]"""
import argparse, ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aline_amd import Aline, Embedder, Encoder, OutputHead, _lib  # noqa: E402
from aline_amd.rollout import Rollout  # noqa: E402
from aline_amd.tasks import HiddenLocation  # noqa: E402
_lib.lib.aline_debug_set_flags(_lib.DBG["WIDE_STAMPS"])

PH = ["setup", "key rows->LDS", "K proj", "V proj", "Q proj", "Q frags+resid", "attention", "OUT proj", "LN1", "FFN",
      "LN2", "acq head"]
ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1000)
ap.add_argument("--T", type=int, default=30)
ap.add_argument("--split", action="store_true", help="split chunk phases into compute / barrier wait / issue")
args = ap.parse_args()
dev = torch.device("cuda")
model = Aline(Embedder(2, 1, 256, 1024, 2, "theta"), Encoder(256, 1024, 8, 0.0, 3), OutputHead(2, 1, 256, 1024),
              precision="bf16").to(dev)
model.train()
task = HiddenLocation(device=dev)
ro = Rollout(model, task.sample_batch(args.batch), args.T, select="sample")
off = _lib.lib.aline_debug_stamps_offset(C.byref(ro.m), C.byref(ro.r))
ro.run(); torch.cuda.synchronize()
ro.ws[off:off + 4 * 16 * 8].zero_()
if args.split:
    ro.ws[off + 63 * 8:off + 64 * 8].view(torch.int64).fill_(1)
    PH += ["chunk: barrier wait", "chunk: issue next", "chunk: compute"]
ro.run(); torch.cuda.synchronize()
st = ro.ws[off:off + 4 * 16 * 8].view(torch.int64).reshape(4, 16).cpu()
tot = st.sum(1).float()
print(f"s_memtime ticks per rollout (wave 0..3 of workgroup 0): {tot.tolist()}")
for i, nm in enumerate(PH):
    row = st[:, i].float()
    print(f"{nm:16s} " + "  ".join(f"{v/1e3:9.1f}k ({100*v/t:4.1f}%)" for v, t in zip(row.tolist(), tot.tolist())))

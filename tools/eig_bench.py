#!/usr/bin/env python3
"""HBM roofline of the sequential-EIG kernels: GB/s of the step kernels and of the
logsumexp finalisation at the README evaluation sizes (location: L = 1e6, B = 200; CES: L = 1e6, B = 20).
Algorithmic bytes per (l, b): theta read (dim_theta * 4 B) + S read + write (8 B); finalize: S read (4 B).
Run on the GPU box:  python tools/eig_bench.py  -> one JSON line."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from aline_amd.loss.eig import EIGStepLoss  # noqa: E402
from aline_amd.tasks import CESTask, HiddenLocation  # noqa: E402

PEAK = 8000.0
dev = torch.device("cuda")
out = {"peak_GBps": PEAK}
for name, task, L, B, dth in (("location", HiddenLocation(device=dev), 1_000_000, 200, 2),
                              ("ces", CESTask(device=dev), 1_000_000, 20, 5)):
    torch.manual_seed(0)
    if name == "location":
        theta = torch.rand(L + 1, B, 1, 2, device=dev)
        xi, y = torch.rand(B, 2, device=dev), torch.randn(B, 1, device=dev)
    else:
        theta = torch.stack([0.01 + 0.99 * torch.rand(L + 1, B, device=dev), *(torch.rand(3, L + 1, B, device=dev) / 3 + 0.1),
                             torch.randn(L + 1, B, device=dev)], -1).contiguous()
        xi, y = torch.rand(B, 6, device=dev) * 100, torch.rand(B, 1, device=dev) * 0.9 + 0.05
    crit = EIGStepLoss(L, B, task, device=dev)
    for _ in range(2): crit.step(y, xi, theta)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 10
    for _ in range(n): crit.step(y, xi, theta)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    step_bytes = (L + 1) * B * (dth * 4 + 8)
    crit.forward(y, xi, theta); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): crit.forward(y, xi, theta)
    torch.cuda.synchronize(); dtf = (time.perf_counter() - t0) / n - dt
    out[name] = {"L": L, "B": B, "step_ms": dt * 1e3, "step_GBps": step_bytes / dt / 1e9, "step_frac_of_hbm_peak": step_bytes / dt / 1e9 / PEAK,
                 "finalize_ms": dtf * 1e3, "finalize_GBps": (L + 1) * B * 4 / max(dtf, 1e-9) / 1e9}
    del theta, crit
    torch.cuda.empty_cache()
print(json.dumps(out))

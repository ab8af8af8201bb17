"""Randomized cross-check of the fused d=32 rollout kernel against the generic fp32 pipeline (same C ABI, env
ALINE_DBG_DISABLE_FUSED): shapes, step counts, target masks, sampling; plus run-to-run reproducibility.
Run on the GPU box:  python tools/fused_sweep.py"""
import sys, os, torch, random
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/oracle")
from helpers import native_model
from aline_amd import _lib
from aline_amd.rollout import Rollout
from aline_amd.tasks import HiddenLocation
DIMS = {"dim_x": 2, "dim_y": 1, "d": 32, "F": 128, "n_head": 4, "L": 3, "C": 10, "n_theta": 2, "embedding_type": "theta", "time_token": False}
def run(env, B, nq, T, seed, mask):
    _lib.lib.aline_debug_set_flags(0); ctx = _lib.debug_env(env); ctx.__enter__()
    model, _ = native_model(DIMS, 3 + seed, "f32")
    torch.manual_seed(seed)
    batch = HiddenLocation(device=torch.device("cuda"), n_query_init=nq).sample_batch(B)
    if mask is not None: batch.target_mask = torch.tensor(mask, dtype=torch.bool, device="cuda")
    g = torch.Generator(device="cpu").manual_seed(seed)
    forced = torch.stack([torch.stack([torch.randint(0, nq - t, (1,), generator=g)[0] for t in range(T)]) for _ in range(B)]).to("cuda")
    ro = Rollout(model, batch, T, select="forced", forced_idx=forced).run()
    torch.cuda.synchronize()
    return ro.target_ll.float().cpu().clone(), ro.log_prob.float().cpu().clone()
random.seed(1)
worst = (0.0, 0.0)
for it in range(40):
    nq = random.choice([5, 13, 14, 29, 45, 61, 100, 157, 200, 220, 236]); T = random.randint(1, min(nq - 1, 30)); B = random.choice([1, 2, 3, 5, 9, 64, 257])
    mask = random.choice([None, None, [True, False], [False, True], [False, False]])
    a = run({}, B, nq, T, it, mask); a2 = run({}, B, nq, T, it, mask); b = run({"ALINE_DISABLE_FUSED": "1"}, B, nq, T, it, mask)
    d1 = float((a[0] - b[0]).abs().max()); d2 = float((a[1] - b[1]).abs().max()); dd = float((a[0] - a2[0]).abs().max()) + float((a[1] - a2[1]).abs().max())
    bad = (not torch.isfinite(a[0]).all()) or d1 > 1e-4 or d2 > 1e-4 or dd != 0.0
    worst = (max(worst[0], d1), max(worst[1], d2))
    print(it, "N", nq + 3, "T", T, "B", B, "mask", mask, "dLL %.2e dlp %.2e rerun %.1e" % (d1, d2, dd), "BAD" if bad else "", flush=True)
print("worst dLL %.2e dlp %.2e" % worst)

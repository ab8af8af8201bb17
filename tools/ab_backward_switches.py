"""A/B of the backward's switches at one shape: max-abs and L2 error of every parameter gradient against the all-switches-off run.
   python tools/ab_backward_switches.py d F H B T tc"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from aline_amd import Aline, Embedder, Encoder, OutputHead, _lib
from aline_amd.rollout import Rollout
from aline_amd.tasks import HiddenLocation
from aline_amd.train import backward, reinforce_terms
d, F, H, B, T, tc = (int(x) for x in sys.argv[1:7])
torch.manual_seed(d + T)
model = Aline(Embedder(2, 1, d, F, 2, "theta"), Encoder(d, F, H, 0.0, 2), OutputHead(2, 1, d, F)).cuda().set_precision("f16x3").train()
with torch.no_grad():
    for p in model.parameters():
        p.add_(0.02 * torch.randn_like(p))
batch = HiddenLocation(n_query_init=45).sample_batch(B)
grads = []
with torch.no_grad():
    ro = Rollout(model, batch, T, select="sample").run()
    terms = reinforce_terms(ro, "theta", "all")
    for flags in ([], ["NO_BWD_KV_SPARSE"], ["NO_BWD_IMAGE_RECOMPUTE", "NO_BWD_KV_SPARSE"], ["NO_BWD_IMAGE_RECOMPUTE", "NO_BWD_KV_SPARSE", "BWD_GRAD_F32"],
                  ["NO_BWD_IMAGE_RECOMPUTE", "NO_BWD_KV_SPARSE", "BWD_GRAD_F32", "BWD_RECOMPUTE_F32"]):
        with _lib.debug(*flags):
            for p in model.parameters():
                p.grad = None
            backward(model, ro, terms["g_logp"], terms["g_ll"], t_chunk=tc)
            torch.cuda.synchronize()
        grads.append({k: p.grad.clone() for k, p in model.named_parameters()})
ref = grads[-1]
floor = 1e-2 * max(float(g.abs().max()) for g in ref.values())
for which in range(len(grads) - 1):
    wm, wl = ("", 0.0), ("", 0.0)
    for k in ref:
        e = float((grads[which][k] - ref[k]).abs().max()) / max(float(ref[k].abs().max()), floor)
        l2 = float((grads[which][k] - ref[k]).norm()) / max(float(ref[k].norm()), 1e-30)
        if e > wm[1]: wm = (k, e)
        if l2 > wl[1]: wl = (k, l2)
    print(which, "max-abs", wm, "L2", wl)

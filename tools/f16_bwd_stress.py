"""Head room of the f16 fused backward kernels: gradients of the d = 32 model with inflated weights / LayerNorm gains, f16 kernels against the
exact-fp32 fused kernels (ALINE_DBG_BWD_GRAD_F32).  python tools/f16_bwd_stress.py"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from aline_amd import Aline, Embedder, Encoder, OutputHead, _lib
from aline_amd.rollout import Rollout
from aline_amd.tasks import HiddenLocation
from aline_amd.train import backward, reinforce_terms
dev = torch.device("cuda")
for wmul, gmul in ((1.0, 1.0), (3.0, 1.0), (1.0, 8.0), (3.0, 8.0), (6.0, 20.0)):
    torch.manual_seed(1)
    model = Aline(Embedder(2, 1, 32, 128, 2, "theta"), Encoder(32, 128, 4, 0.0, 3), OutputHead(2, 1, 32, 128)).cuda().set_precision("f16x3").train()
    with torch.no_grad():
        for n, p in model.named_parameters():
            if "norm" in n and n.endswith("weight"):
                p.mul_(gmul)
            elif p.dim() > 1:
                p.mul_(wmul)
    batch = HiddenLocation(n_query_init=100, device=dev).sample_batch(64)
    grads = []
    with torch.no_grad():
        ro = Rollout(model, batch, 20, select="sample").run()
        st = ro.range_status()
        terms = reinforce_terms(ro, "theta", "all")
        for flags in ([], ["BWD_GRAD_F32"]):
            with _lib.debug(*flags):
                for p in model.parameters():
                    p.grad = None
                backward(model, ro, terms["g_logp"], terms["g_ll"])
                torch.cuda.synchronize()
            grads.append({k: p.grad.clone() for k, p in model.named_parameters()})
    new, ref = grads
    floor = 1e-2 * max(float(g.abs().max()) for g in ref.values())
    finite = all(bool(torch.isfinite(v).all()) for v in new.values())
    errs = sorted(((float((new[k] - ref[k]).abs().max()) / max(float(ref[k].abs().max()), floor), k) for k in ref), reverse=True)
    print(f"weights x{wmul} LN gains x{gmul}: rollout status {st}, finite {finite}, max |grad| {max(float(g.abs().max()) for g in ref.values()):.3g}, worst {errs[0][0]:.2e} ({errs[0][1]}), median {errs[len(errs) // 2][0]:.2e}")

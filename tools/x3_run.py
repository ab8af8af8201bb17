"""Two eager rollouts of the x3 path at the headline shape (d = 256 / F = 1024 / 8 heads, f16x3, B = 1000): a minimal target for
rocprofv3 --pmc passes.    python tools/x3_run.py [T]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from aline_amd import Aline, Embedder, Encoder, OutputHead
from aline_amd.rollout import Rollout
from aline_amd.tasks import HiddenLocation
torch.manual_seed(0)
dev = torch.device("cuda")
T = int(sys.argv[1]) if len(sys.argv) > 1 else 30
m = Aline(Embedder(2, 1, 256, 1024, 2, "theta"), Encoder(256, 1024, 8, 0.0, 3), OutputHead(2, 1, 256, 1024)).cuda().set_precision("f16x3").train()
batch = HiddenLocation(n_query_init=200, device=dev).sample_batch(1000)
ro = Rollout(m, batch, T, select="sample", keep_posterior=True)
assert ro.path == "x3::layer_kernel", ro.path
ro.run(); torch.cuda.synchronize()
ro.refresh_uniform(); ro.run(); torch.cuda.synchronize()
print("ok", float(ro.target_ll.mean()), ro.range_status())

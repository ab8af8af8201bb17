"""Two eager rollouts of the x5 path at the bench's d512 leg (d = 512 / F = 128 / 8 heads of 64, f16x3, B = 1000, T = 30, n_query = 200):
a minimal target for rocprofv3 --pmc passes.    python tools/d512_run.py [T]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from aline_amd import Aline, Embedder, Encoder, OutputHead
from aline_amd.rollout import Rollout
from aline_amd.tasks import HiddenLocation
torch.manual_seed(0)
dev = torch.device("cuda")
T = int(sys.argv[1]) if len(sys.argv) > 1 else 30
m = Aline(Embedder(2, 1, 512, 128, 2, "theta"), Encoder(512, 128, 8, 0.0, 3), OutputHead(2, 1, 512, 128)).cuda().set_precision("f16x3").train()
batch = HiddenLocation(n_query_init=200, device=dev).sample_batch(1000)
ro = Rollout(m, batch, T, select="sample", keep_posterior=True)
assert ro.path == "x5::layer_kernel", ro.path
ro.run(); torch.cuda.synchronize()
ro.refresh_uniform(); ro.run(); torch.cuda.synchronize()
print("ok", float(ro.target_ll.mean()), ro.range_status())

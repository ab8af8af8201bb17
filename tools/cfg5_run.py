"""A few eager rollouts of cfg5 (psychometric, d = 512, 8 heads, predefined mask) for rocprofv3 --kernel-trace --stats.
    python tools/cfg5_run.py [f16x3|bf16] [F]"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from aline_amd import Aline, Embedder, Encoder, OutputHead
from aline_amd.rollout import Rollout
from aline_amd.tasks import PsychometricTask
torch.manual_seed(0)
dev = torch.device("cuda")
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 128
m = Aline(Embedder(1, 1, 512, F, 4, "theta"), Encoder(512, F, 8, 0.0, 3), OutputHead(1, 1, 512, F)).cuda().set_precision(prec).train()
batch = PsychometricTask(n_query_init=200, n_context_init=1, device=dev).sample_batch(256)
batch["target_mask"] = torch.tensor([False, False, True, True])
ro = Rollout(m, batch, 30, select="sample", keep_posterior=True)
print(ro.path)
for _ in range(3):
    ro.refresh_uniform(); ro.run(); torch.cuda.synchronize()

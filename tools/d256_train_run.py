"""One warm + one profiled training step of the d = 256 / F = 1024 / 8 heads model at the headline batch (for rocprofv3 --kernel-trace --stats)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from aline_amd import Aline, Embedder, Encoder, OutputHead
from aline_amd.tasks import HiddenLocation
from aline_amd.train import train_step
torch.manual_seed(0)
dev = torch.device("cuda")
d, F = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 1024)
m = Aline(Embedder(2, 1, d, F, 2, "theta"), Encoder(d, F, 8, 0.0, 3), OutputHead(2, 1, d, F)).cuda().set_precision("f16x3").train()
batch = HiddenLocation(n_query_init=200, device=dev).sample_batch(1000)
opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
for _ in range(2):
    train_step(m, batch, 30, optimizer=opt)
    torch.cuda.synchronize()

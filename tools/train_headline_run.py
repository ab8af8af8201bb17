import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from aline_amd import Aline, Embedder, Encoder, OutputHead
from aline_amd.tasks import HiddenLocation
from aline_amd.train import train_step
torch.manual_seed(0)
dev = torch.device("cuda")
m = Aline(Embedder(2, 1, 32, 128, 2, "theta"), Encoder(32, 128, 4, 0.0, 3), OutputHead(2, 1, 32, 128)).cuda().set_precision("f16x3").train()
batch = HiddenLocation(n_query_init=200, device=dev).sample_batch(1000)
opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
for _ in range(3):
    train_step(m, batch, 30, optimizer=opt)
    torch.cuda.synchronize()

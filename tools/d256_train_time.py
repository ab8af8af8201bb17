"""Time the training step of the d = 256 / F = 1024 model (bench `d256.train_step` shape) under debug flags: python tools/d256_train_time.py [FLAGS ...]"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from aline_amd import Aline, Embedder, Encoder, OutputHead, _lib
from aline_amd.tasks import HiddenLocation
from aline_amd.train import train_step
torch.manual_seed(0)
dev = torch.device("cuda")
d, F = 256, 1024
m = Aline(Embedder(2, 1, d, F, 2, "theta"), Encoder(d, F, 8, 0.0, 3), OutputHead(2, 1, d, F)).cuda().set_precision("f16x3").train()
batch = HiddenLocation(n_query_init=200, device=dev).sample_batch(1000)
opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
with _lib.debug(*sys.argv[1:]):
    train_step(m, batch, 30, optimizer=opt); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        terms, _ = train_step(m, batch, 30, optimizer=opt)
    torch.cuda.synchronize()
    print(sys.argv[1:], "d256 train step %.1f ms" % ((time.perf_counter() - t0) / 2 * 1e3), "loss", float(terms["loss"]))

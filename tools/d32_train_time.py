"""Time the training step at the bench shape (location_finding, B = 1000, T = 30, n_query = 200, default d = 32 model): python tools/d32_train_time.py [FLAGS ...]"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from aline_amd import Aline, Embedder, Encoder, OutputHead, _lib
from aline_amd.tasks import HiddenLocation
from aline_amd.train import train_step
torch.manual_seed(0)
dev = torch.device("cuda")
m = Aline(Embedder(2, 1, 32, 128, 2, "theta"), Encoder(32, 128, 4, 0.0, 3), OutputHead(2, 1, 32, 128)).cuda().set_precision("f16x3").train()
batch = HiddenLocation(n_query_init=200, device=dev).sample_batch(1000)
opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
with _lib.debug(*sys.argv[1:]):
    for _ in range(3):
        train_step(m, batch, 30, optimizer=opt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        terms, _ = train_step(m, batch, 30, optimizer=opt)
    torch.cuda.synchronize()
    print(sys.argv[1:], "d32 train step %.2f ms" % ((time.perf_counter() - t0) / 10 * 1e3), "loss", float(terms["loss"]))

"""Diagnostic: the first words of a rollout workspace (status word of the f16 range guard at byte 0) after eager runs and graph replays."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from aline_amd import Aline, Embedder, Encoder, OutputHead
from aline_amd.rollout import Rollout
from aline_amd.tasks import HiddenLocation
torch.manual_seed(123)
dev = torch.device("cuda")
m = Aline(Embedder(2, 1, 32, 128, 2, "theta"), Encoder(32, 128, 4, 0.0, 3), OutputHead(2, 1, 32, 128)).cuda().set_precision("f16x3").train()
batch = HiddenLocation(n_query_init=200, device=dev).sample_batch(1000)
ro = Rollout(m, batch, 30, select="sample", keep_zt=False, keep_posterior=True)
def show(tag):
    torch.cuda.synchronize()
    print(tag, ro.ws[:96].view(torch.int32).tolist(), "status", ro.range_status(), flush=True)
show("fresh")
ro.run(); show("eager 1")
ro.run(); show("eager 2")
ro.capture(); show("captured")
for i in range(3):
    ro.refresh_uniform(); ro.replay(); show(f"replay {i}")
for i in range(200):
    ro.refresh_uniform(); ro.replay()
show("after 200 replays")
x = torch.ones(1, device=dev); y = x * 2
ro.replay(); show("after alloc + replay")

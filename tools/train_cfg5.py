"""Training step at the cfg5 shape (psychometric, d = 512 / F = 128 / 8 heads of 64, predefined mask, B = 256, T = 30): x5 rollout,
per-op backward (no fused backward kernels at this width).  Prints ms per optimiser step."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from aline_amd import Aline, Embedder, Encoder, OutputHead
from aline_amd.tasks import PsychometricTask
from aline_amd.train import train_step
torch.manual_seed(0)
dev = torch.device("cuda")
m = Aline(Embedder(1, 1, 512, 128, 4, "theta"), Encoder(512, 128, 8, 0.0, 3), OutputHead(1, 1, 512, 128)).cuda().set_precision("f16x3")
batch = PsychometricTask(n_query_init=200, n_context_init=1, device=dev).sample_batch(256)
batch["target_mask"] = torch.tensor([False, False, True, True])
opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
train_step(m, batch, 30, optimizer=opt, mask_type="predefined"); torch.cuda.synchronize()
t0 = time.time()
for _ in range(2):
    terms, ro = train_step(m, batch, 30, optimizer=opt, mask_type="predefined")
torch.cuda.synchronize()
print(f"cfg5 train step: {(time.time() - t0) / 2 * 1e3:.1f} ms, loss {float(terms['loss']):.4f}, path {ro.path}, peak GB {torch.cuda.max_memory_allocated() / 1e9:.1f}")

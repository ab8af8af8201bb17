"""Training step at the cfg3 shape (al_mix dx=2, B=512 = one GPU's share of 4096, T=50, n_query=200, split mask (data), d=32):
the 8-GPU config of BASELINE.json.  Up to 151 keys per instance: the per-op attention kernels with the fused tail /
acquisition / GMM kernels; t_chunk from the 24 GB workspace cap.  Prints ms per optimiser step."""
import sys, time, torch
sys.path.insert(0, "/root/repo")
from aline_amd import Aline, Embedder, Encoder, OutputHead
from aline_amd.tasks import GPTask
from aline_amd.train import train_step
from aline_amd.utils import create_target_mask
dev = torch.device("cuda")
torch.manual_seed(0)
task = GPTask(dim_x=2, embedding_type="mix", n_context_init=1, n_query_init=200, n_target_theta=3, n_target_data=100, device=dev)
batch = task.sample_batch(int(sys.argv[1]) if len(sys.argv) > 1 else 512)
batch["target_mask"] = create_target_mask("split", "mix", 100, 3, None, None, None, None, "data")
m = Aline(Embedder(2, 1, 32, 128, 3, "mix"), Encoder(32, 128, 4, 0.0, 3), OutputHead(2, 1, 32, 128)).cuda().set_precision("f16x3")
opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
T = 50
train_step(m, batch, T, optimizer=opt, embedding_type="mix", mask_type="split"); torch.cuda.synchronize()
t0 = time.time()
for _ in range(2):
    terms, ro = train_step(m, batch, T, optimizer=opt, embedding_type="mix", mask_type="split")
torch.cuda.synchronize()
dt = (time.time() - t0) / 2
print(f"cfg3 train step: {dt * 1e3:.1f} ms, loss {float(terms['loss']):.4f}, path {ro.path}, ws GB {torch.cuda.max_memory_allocated() / 1e9:.1f}")

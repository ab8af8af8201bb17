"""Graph-replay time of the s3 rollout at the headline (2) or cfg3 (3) shape: python tools/s3_time.py [2|3] [replays] [FLAGS ...]"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from aline_amd import Aline, Embedder, Encoder, OutputHead, _lib
from aline_amd.rollout import Rollout
from aline_amd.tasks import GPTask, HiddenLocation
from aline_amd.utils import create_target_mask
torch.manual_seed(0)
dev = torch.device("cuda")
cfg = sys.argv[1] if len(sys.argv) > 1 else "2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
if cfg == "3":
    m = Aline(Embedder(2, 1, 32, 128, 3, "mix"), Encoder(32, 128, 4, 0.0, 3), OutputHead(2, 1, 32, 128))
    batch = GPTask(dim_x=2, embedding_type="mix", n_context_init=1, n_query_init=200, n_target_theta=3, n_target_data=100, device=dev).sample_batch(512)
    batch["target_mask"] = create_target_mask("split", "mix", 100, 3, None, None, None, None, "data")
    T = 50
else:
    m = Aline(Embedder(2, 1, 32, 128, 2, "theta"), Encoder(32, 128, 4, 0.0, 3), OutputHead(2, 1, 32, 128))
    batch = HiddenLocation(n_query_init=200, device=dev).sample_batch(1000)
    T = 30
m = m.cuda().set_precision("f16x3").train()
with _lib.debug(*sys.argv[3:]):
    ro = Rollout(m, batch, T, select="sample", keep_zt=False, keep_posterior=True)
    assert ro.path == "s3::step_kernel", ro.path
    ro.run(); torch.cuda.synchronize()
    ro.capture()
    for _ in range(100):
        ro.refresh_uniform(); ro.replay()
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(n):
            ro.refresh_uniform(); ro.replay()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n)
    print(sys.argv[3:], os.environ.get("ALINE_HIP_LIB", "default lib").split("/")[-1], "cfg", cfg, "%.4f ms per rollout" % (best * 1e3), "nll", -float(ro.target_ll.mean()), "idx sum", int(ro.idx.sum()))

"""Timing of the x3 path (d = 256 / F = 1024 / 8 heads, f16x3) at the headline shape: ms per graph-replayed rollout, the layer
kernel's launch time (HIP events around the last layer of the last step) and its matrix-pipe fraction.  Used for same-box A/B
runs of library variants:  ALINE_HIP_LIB=aline_amd/csrc/variants/lib_NAME.so python tools/x3_time.py [B] [T] [reps]"""
import os, sys, time, json, torch
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, R)
from aline_amd import Aline, Embedder, Encoder, OutputHead
from aline_amd.rollout import Rollout
from aline_amd.tasks import HiddenLocation
sys.path.insert(0, R)
from bench import HipEvents, x3_layer_flops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
T = int(sys.argv[2]) if len(sys.argv) > 2 else 30
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
torch.manual_seed(0)
dev = torch.device("cuda")
m = Aline(Embedder(2, 1, 256, 1024, 2, "theta"), Encoder(256, 1024, 8, 0.0, 3), OutputHead(2, 1, 256, 1024)).cuda().set_precision("f16x3").train()
batch = HiddenLocation(n_query_init=200, device=dev).sample_batch(B)
ro = Rollout(m, batch, T, select="sample", keep_posterior=True)
assert ro.path == "x3::layer_kernel", ro.path
ro.run(); torch.cuda.synchronize()
ro.capture(); ro.refresh_uniform(); ro.replay(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    ro.refresh_uniform(); ro.replay()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / reps * 1e3
ev = HipEvents()
ro.r.ev_kernel_start, ro.r.ev_kernel_stop = ev.a, ev.b
ks = []
for _ in range(3):
    ro.refresh_uniform(); ro.run(); torch.cuda.synchronize()
    ks.append(ev.elapsed_ms())
k = sum(ks) / len(ks)
fl = x3_layer_flops(256, 1024, 1 + T - 1, 200 - (T - 1), 2, 2) * B
print(json.dumps({"lib": os.path.basename(os.environ.get("ALINE_HIP_LIB", "libaline_hip.so")), "B": B, "T": T, "ms_per_rollout": round(ms, 3),
                  "layer_kernel_us": round(k * 1e3, 1), "frac_2.5PF": round(fl / (k * 1e-3) / 2.5e15, 4), "pipe_frac": round(3 * fl / (k * 1e-3) / 2.5e15, 4),
                  "ll_mean": float(ro.target_ll.mean()), "range": ro.range_status()}))

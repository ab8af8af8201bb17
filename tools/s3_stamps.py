"""In-kernel phase stamps of s3::step_kernel (diagnostic S3_STAMPS build, ALINE_HIP_LIB=.../lib_s3stamps.so): where the
waves of workgroup 0 spend their cycles in the LAST step launch of a cfg3-shaped (B=512, al_mix, split mask) or cfg2-shaped
(headline: location finding, B=1000) rollout.
    tools/x3_variants.sh build "s3stamps:-DS3_STAMPS";  ALINE_HIP_LIB=aline_amd/csrc/variants/lib_s3stamps.so python tools/s3_stamps.py [T [2|3]]"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from aline_amd import Aline, Embedder, Encoder, OutputHead, _lib
from aline_amd.rollout import Rollout
from aline_amd.tasks import GPTask, HiddenLocation
from aline_amd.utils import create_target_mask
torch.manual_seed(0)
dev = torch.device("cuda")
cfg = sys.argv[2] if len(sys.argv) > 2 else "3"
if cfg == "3":      # cfg3: al_mix, B = 512, 304 tokens, up to 150 keys
    m = Aline(Embedder(2, 1, 32, 128, 3, "mix"), Encoder(32, 128, 4, 0.0, 3), OutputHead(2, 1, 32, 128))
    task = GPTask(dim_x=2, embedding_type="mix", n_context_init=1, n_query_init=200, n_target_theta=3, n_target_data=100, device=dev)
    batch = task.sample_batch(512)
    batch["target_mask"] = create_target_mask("split", "mix", 100, 3, None, None, None, None, "data")
else:               # cfg2 (headline): location finding, B = 1000, 203 tokens, up to 32 keys
    m = Aline(Embedder(2, 1, 32, 128, 2, "theta"), Encoder(32, 128, 4, 0.0, 3), OutputHead(2, 1, 32, 128))
    batch = HiddenLocation(n_query_init=200, device=dev).sample_batch(1000)
m = m.cuda().set_precision("f16x3").train()
T = int(sys.argv[1]) if len(sys.argv) > 1 else (50 if cfg == "3" else 30)
ro = Rollout(m, batch, T, select="sample", keep_posterior=False)
ro.r.target_ll = None
assert ro.path == "s3::step_kernel", ro.path
ro.run(); torch.cuda.synchronize(); ro.run(); torch.cuda.synchronize()
off = _lib.lib.aline_debug_xraw_offset(C.byref(ro.m), C.byref(ro.r))
st = ro.ws[off:off + 8 * 9 * 8].view(torch.int64).reshape(8, 9).cpu().double()
names = ["prologue", "K/V tiles", "barrier (K/V)", "load + Q", "attention", "out-proj..LN2+head", "barrier (tiles)", "next weights", "TOTAL"]
print(f"cycles per wave (workgroup 0), step {T - 1}; share of the wave's total:")
for w in range(8):
    tot = st[w, 8]
    print(f"wave {w}: " + "  ".join(f"{names[k]} {st[w, k] / tot * 100:5.1f}%" for k in range(8)) + f"   total {tot / 1e3:.0f} kcyc")

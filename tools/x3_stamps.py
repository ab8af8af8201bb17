"""In-kernel phase stamps of x3::layer_kernel (diagnostic X3_STAMPS build, ALINE_HIP_LIB=.../lib_stamps.so): where the
waves of workgroup 0 spend their cycles in the LAST layer launch of a T-step rollout at the headline shape.
    tools/x3_variants.sh build "stamps:-DX3_STAMPS";  ALINE_HIP_LIB=aline_amd/csrc/variants/lib_stamps.so python tools/x3_stamps.py [T] [d] [F] [B]"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from aline_amd import Aline, Embedder, Encoder, OutputHead, _lib
from aline_amd.rollout import Rollout
from aline_amd.tasks import HiddenLocation
torch.manual_seed(0)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 4
d = int(sys.argv[2]) if len(sys.argv) > 2 else 256          # 256: x3 (8 waves per workgroup), 512: x5 (4 waves)
F = int(sys.argv[3]) if len(sys.argv) > 3 else (1024 if d == 256 else 128)
B = int(sys.argv[4]) if len(sys.argv) > 4 else (1000 if d == 256 else 256)
m = Aline(Embedder(2, 1, d, F, 2, "theta"), Encoder(d, F, 8, 0.0, 3), OutputHead(2, 1, d, F)).cuda().set_precision("f16x3").train()
batch = HiddenLocation(n_query_init=200, device=torch.device("cuda")).sample_batch(B)
ro = Rollout(m, batch, T, select="sample", keep_posterior=False)
ro.r.target_ll = None
ro.run(); torch.cuda.synchronize(); ro.run(); torch.cuda.synchronize()
off = _lib.lib.aline_debug_xraw_offset(C.byref(ro.m), C.byref(ro.r))
print(ro.path)
NW = 8 if d == 256 else 4
st = ro.ws[off:off + NW * 9 * 8].view(torch.int64).reshape(NW, 9).cpu().double()
names = ["LDS wait", "MFMA batch", "DMA issue", "vmcnt wait", "barrier", "hidden/epilogue VALU", "attention", "LN + image I/O", "TOTAL"]
print("cycles per wave (workgroup 0), last layer launch; share of the wave's total:")
for w in range(NW):
    tot = st[w, 8]
    print(f"wave {w}: " + "  ".join(f"{names[k]} {st[w, k] / tot * 100:5.1f}%" for k in range(8)) + f"   total {tot / 1e3:.0f} kcyc")

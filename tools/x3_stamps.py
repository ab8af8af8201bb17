"""In-kernel phase stamps of x3::layer_kernel (diagnostic X3_STAMPS build, ALINE_HIP_LIB=.../lib_stamps.so): where the
waves of workgroup 0 spend their cycles in the LAST layer launch of a T-step rollout at the headline shape.
    tools/x3_variants.sh build "stamps:-DX3_STAMPS";  ALINE_HIP_LIB=aline_amd/csrc/variants/lib_stamps.so python tools/x3_stamps.py"""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from aline_amd import Aline, Embedder, Encoder, OutputHead, _lib
from aline_amd.rollout import Rollout
from aline_amd.tasks import HiddenLocation
torch.manual_seed(0)
m = Aline(Embedder(2, 1, 256, 1024, 2, "theta"), Encoder(256, 1024, 8, 0.0, 3), OutputHead(2, 1, 256, 1024)).cuda().set_precision("f16x3").train()
batch = HiddenLocation(n_query_init=200, device=torch.device("cuda")).sample_batch(1000)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ro = Rollout(m, batch, T, select="sample", keep_posterior=False)
ro.r.target_ll = None
ro.run(); torch.cuda.synchronize(); ro.run(); torch.cuda.synchronize()
off = _lib.lib.aline_debug_xraw_offset(C.byref(ro.m), C.byref(ro.r))
st = ro.ws[off:off + 8 * 9 * 8].view(torch.int64).reshape(8, 9).cpu().double()
names = ["LDS wait", "MFMA batch", "DMA issue", "vmcnt wait", "barrier", "hidden/epilogue VALU", "attention", "LN + image I/O", "TOTAL"]
print("cycles per wave (workgroup 0), last layer launch; share of the wave's total:")
for w in range(8):
    tot = st[w, 8]
    print(f"wave {w}: " + "  ".join(f"{names[k]} {st[w, k] / tot * 100:5.1f}%" for k in range(8)) + f"   total {tot / 1e3:.0f} kcyc")

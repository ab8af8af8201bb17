"""Wide path (d=256, bf16) accuracy probe against the committed reference fixture."""
import sys, os, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/oracle")
from conftest import Fixture
from helpers import native_model, to_dev
from aline_amd import _lib
from aline_amd.rollout import Rollout
fx = Fixture("cfg2_location_d256")
T = fx.meta["T"]
ref_ll = torch.stack([fx.t(f"train.target_ll_{t}") for t in range(T)])
outs = {}
for name, prec, env in (("bf16 wide-step", "bf16", {}), ("f32 generic", "f32", {}), ("bf16 generic", "bf16", {"ALINE_DISABLE_WIDE": "1"}),
                        ("bf16 wide-blocks", "bf16", {"ALINE_WIDE_BLOCKS": "1"})):
    _lib.lib.aline_debug_set_flags(0); _lib.debug_env(env).__enter__()
    model, _ = native_model(fx.meta["dims"], fx.meta["wseed"], prec)
    ro = Rollout(model, to_dev(fx.batch()), T, select="forced", forced_idx=fx.forced_idx("train"), keep_zt=True).run()
    torch.cuda.synchronize()
    zt_ref = fx.t("train.zt_0")
    outs[name] = (ro.target_ll.cpu().clone(), ro.log_prob.cpu().clone())
    print(f"{name:18s} max|dLL|={float((ro.target_ll.cpu() - ref_ll).abs().max()):.6f}",
          f"max|dlogp|={float((ro.log_prob.cpu() - fx.t('train.log_probs')).abs().max()):.6f}",
          f"max|dzt0|={float((ro.zt[0].cpu()[:, :zt_ref.shape[1]] - zt_ref).abs().max()):.6f}", flush=True)
_lib.lib.aline_debug_set_flags(0)
a, b = outs["bf16 wide-blocks"], outs["bf16 wide-step"]
print("step vs blocks: max|dLL|=%.6f max|dlogp|=%.6f" % (float((a[0] - b[0]).abs().max()), float((a[1] - b[1]).abs().max())))

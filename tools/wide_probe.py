import sys, os, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/oracle")
from conftest import Fixture
from helpers import native_model, to_dev
from aline_amd.rollout import Rollout
fx = Fixture("cfg2_location_d256")
T = fx.meta["T"]
ref_ll = torch.stack([fx.t(f"train.target_ll_{t}") for t in range(T)])
for prec, env in (("f32", None), ("bf16", "ALINE_DISABLE_WIDE"), ("bf16", None)):
    if env: os.environ[env] = "1"
    else: os.environ.pop("ALINE_DISABLE_WIDE", None)
    model, _ = native_model(fx.meta["dims"], fx.meta["wseed"], prec)
    ro = Rollout(model, to_dev(fx.batch()), T, select="forced", forced_idx=fx.forced_idx("train"), keep_zt=True).run()
    torch.cuda.synchronize()
    zt_ref = fx.t("train.zt_0")
    print(prec, "wide" if (prec == "bf16" and not env) else "generic",
          "max|dLL|=%.4f" % float((ro.target_ll.cpu() - ref_ll).abs().max()),
          "max|dlogp|=%.4f" % float((ro.log_prob.cpu() - fx.t("train.log_probs")).abs().max()),
          "max|dzt0|=%.5f" % float((ro.zt[0].cpu()[:, :zt_ref.shape[1]] - zt_ref).abs().max()))
os.environ.pop("ALINE_DISABLE_WIDE", None)

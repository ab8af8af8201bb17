"""Error of the CES log-likelihood kernel against the reference fixture (tests/golden/eig): table kernel vs the
generic one (ALINE_CES_GENERIC=1 in the environment selects it).   python tools/ces_parity.py"""
import os
import sys

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
for p in ("", "tests", "oracle"):
    sys.path.insert(0, os.path.join(ROOT, p))
from conftest import Fixture  # noqa: E402
from aline_amd.tasks import CESTask  # noqa: E402

fx = Fixture("eig")
task = CESTask()
th0, x, y, th = (fx.t(k).cuda() for k in ("ces_theta0", "ces_x", "ces_y", "ces_thetas"))
thetas = torch.cat([th0.unsqueeze(0), th], 0).contiguous()
ref = fx.t("ces_ll")
worst_abs = worst_rel = 0.0
for t in range(x.shape[1]):
    ll = task.log_likelihood(y[:, t].unsqueeze(0), x[:, t].unsqueeze(0), thetas).cpu()
    fin = torch.isfinite(ref[t])
    assert (torch.isfinite(ll) == fin).all()
    d = (ll[fin] - ref[t][fin]).abs()
    worst_abs = max(worst_abs, float(d.max()))
    worst_rel = max(worst_rel, float((d / (ref[t][fin].abs() + 1.0)).max()))
print(f"kernel={'generic' if 'CES_GENERIC' in os.environ.get('ALINE_DBG', '') else 'table'} max|d|={worst_abs:.4g} "
      f"max|d|/(|ref|+1)={worst_rel:.3g}  (|ref| max {float(ref[torch.isfinite(ref)].abs().max()):.3g})")

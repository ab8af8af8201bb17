"""GPU: EIGStepLoss / compute_EIG_from_history kernels vs the reference golden vectors (a14-a16)."""
import math

import pytest
import torch

import aline_oracle as orc
from helpers import maxdiff

pytestmark = pytest.mark.gpu


def test_location_eig_bounds_match_reference(golden):
    from aline_amd.loss import EIGStepLoss
    from aline_amd.tasks import HiddenLocation
    fx = golden("eig")
    task = HiddenLocation()
    th0, x, y, th = (fx.t(k).cuda() for k in ("loc_theta0", "loc_x", "loc_y", "loc_thetas"))
    thetas = torch.cat([th0.unsqueeze(0), th], 0).contiguous()
    ll = task.log_likelihood(y[:, 0].unsqueeze(0), x[:, 0].unsqueeze(0), thetas)
    assert maxdiff(ll, fx.t("loc_ll_step0")) < 1e-4
    L, B = th.shape[0], th.shape[1]
    crit = EIGStepLoss(L, B, task, reduction="none")
    for t in range(x.shape[1]):
        pce, nmc = crit(y[:, t], x[:, t], thetas)
        assert maxdiff(math.log(L + 1) - pce, fx.t("loc_pce")[:, t]) < 2e-4
        assert maxdiff(math.log(L) - nmc, fx.t("loc_nmc")[:, t]) < 2e-4


def test_location_eig_large_L_streaming_property():
    """Full-size shape property: the streaming LSE over L=2e5 equals an fp64 logsumexp."""
    from aline_amd.loss import EIGStepLoss
    from aline_amd.tasks import HiddenLocation
    task = HiddenLocation()
    torch.manual_seed(0)
    L, B, T = 200_000, 50, 3
    th0 = task.sample_theta(B)
    thetas = torch.cat([th0.unsqueeze(0), task.sample_theta((L, B))], 0).contiguous()
    x = torch.rand(B, T, 2, device="cuda")
    y = torch.stack([task.forward(x[:, t], th0) for t in range(T)], 1)
    crit = EIGStepLoss(L, B, task, reduction="none")
    for t in range(T):
        pce, nmc = crit(y[:, t], x[:, t], thetas)
    S = crit.seq_logprobs.double()
    ref_pce = S.logsumexp(0) - S[0]
    ref_nmc = S[1:].logsumexp(0) - S[0]
    assert maxdiff(pce, ref_pce.float().cpu()) < 1e-3
    assert maxdiff(nmc, ref_nmc.float().cpu()) < 1e-3
    # and S itself against the CPU oracle on a slice
    sl = slice(0, 257)
    ref_S = sum(orc.location_log_likelihood(y[:, t].cpu().unsqueeze(0), x[:, t].cpu().unsqueeze(0),
                                            thetas[sl].cpu()).squeeze(-1) for t in range(T))
    assert maxdiff(crit.seq_logprobs[sl], ref_S) < 2e-3

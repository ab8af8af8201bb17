"""CPU: the round-2 fixtures (oracle/make_golden_r2.py, produced by the reference itself) against (1) the oracle and
(2) the product's host-side logic -- target masks incl. their RNG-consuming branches (a13), GP kernel matrices and the
psychometric function (f1), calculate_gmm_variance (f4)."""
import json
import os
import random

import numpy as np
import pytest
import torch

import aline_oracle as orc
from conftest import GOLDEN

DEEP = ["deep_cfg3_almix_d2", "deep_cfg5_psycho_d512", "deep_cfg2_location_d256"]


@pytest.mark.parametrize("name", DEEP)
def test_oracle_matches_deep_rollouts(golden, name):
    """Full-depth rollouts (cfg3: T = 50, every one of the 151 keys becomes visible; cfg5, cfg2 d=256: T = 30)."""
    fx = golden(name)
    dims, T = fx.meta["dims"], fx.meta["T"]
    sd = orc.make_state_dict(fx.meta["wseed"], **dims)
    forced = fx.t("train.idx")
    r = orc.rollout(sd, fx.batch(), fx.cfg(), T, forced_idx=forced, mask_type=fx.meta["mask_type"])
    ll = torch.stack(r["target_ll"])
    assert float((ll - fx.t("train.target_ll")).abs().max()) < 1e-4
    lp = torch.stack(r["log_prob"], 1)
    assert float((lp - fx.t("train.log_probs")).abs().max()) < 1e-4
    R, dl, pl = orc.reinforce_losses(lp, r["nll_q"], r["nll"])
    # R is a z-score of clamped NLL differences: differences of 1e-5 at the clamp threshold move single entries
    assert float((R - fx.t("train.R")).abs().max()) < 5e-3
    assert abs(float(dl) - float(fx.np("train.design_loss"))) < 1e-4
    assert abs(float(pl) - float(fx.np("train.predict_loss"))) < 1e-5
    assert torch.equal(r["batch"]["context_x"], fx.t("train.final_context_x"))


def test_oracle_time_token_eval_schedule(golden):
    """utils/eval.py:24 feeds (T - t) / T; the free-running argmax rollout reproduces the reference's designs."""
    fx = golden("aux_timetoken_eval")
    dims, T = fx.meta["dims"], fx.meta["T"]
    sd = orc.make_state_dict(fx.meta["wseed"], **dims)
    r = orc.rollout(sd, fx.batch(), fx.cfg(), T, forced_idx=fx.t("eval.idx"), time_schedule="eval")
    for t in range(T):
        assert float((r["zt"][t] - fx.t(f"eval.zt_{t}")).abs().max()) < 5e-5
    # and the training schedule t / T is a different function of t
    r2 = orc.rollout(sd, fx.batch(), fx.cfg(), T, forced_idx=fx.t("eval.idx"), time_schedule="train")
    assert float((r2["zt"][0] - fx.t("eval.zt_0")).abs().max()) > 1e-4
    assert torch.equal(fx.t("traces_y"), fx.t("eval.final_context_y"))      # get_traces == the recorded loop


def test_oracle_ces_realistic_regime(golden):
    """CES likelihood where it discriminates (log-likelihoods of a few units, interior and censored outcomes)."""
    fx = golden("eig_r2")
    th0, x, y, th = fx.t("ces_theta0"), fx.t("ces_x"), fx.t("ces_y"), fx.t("ces_thetas")
    th_all = torch.cat([th0.unsqueeze(0), th], 0)
    for t in range(x.shape[1]):
        ll = orc.ces_log_likelihood(y[:, t].unsqueeze(0), x[:, t].unsqueeze(0), th_all)
        assert torch.allclose(ll, fx.t("ces_ll")[t], rtol=1e-4, atol=2e-4), float((ll - fx.t("ces_ll")[t]).abs().max())
    pce, nmc, _ = orc.eig_bounds_from_history(orc.ces_log_likelihood, th0, x, y, th, stepwise=True)
    assert float((pce - fx.t("ces_pce")).abs().max()) < 5e-4 and float((nmc - fx.t("ces_nmc")).abs().max()) < 5e-4


# ---- product host logic -----------------------------------------------------------------------------------------
def test_product_target_mask_known_answers(golden):
    """aline_amd.utils.create_target_mask against the reference's masks (utils/target_mask.py:168-218 known answers)."""
    from aline_amd.utils import create_target_mask, select_targets_by_mask
    fx = golden("masks")
    pre = [[False, False, True, True], [True, True, False, False]]

    def mk(mask_type, emb, n_td, n_th, predefined=None, mask_index=None, attend_to=None):
        return create_target_mask(mask_type, emb, n_td, n_th, None, predefined, None, mask_index, attend_to).tolist()

    assert mk("all", "theta", 0, 4) == fx.np("mask_all_theta").tolist() == [True] * 4
    assert mk("none", "data", 5, 0) == fx.np("mask_none_data").tolist() == [False] * 5
    assert mk("predefined", "theta", 0, 4, pre, 0) == fx.np("mask_predef0").tolist()
    assert mk("predefined", "theta", 0, 4, pre, 1) == fx.np("mask_predef1").tolist()
    assert mk("split", "mix", 5, 3, attend_to="data") == fx.np("mask_split_data").tolist() == [True] * 5 + [False] * 3
    assert mk("split", "mix", 5, 3, attend_to="theta") == fx.np("mask_split_theta").tolist()
    tm = torch.tensor([True, False, False, True, False])
    assert torch.equal(select_targets_by_mask(fx.t("select_in"), tm), fx.t("select_out"))


def test_product_target_mask_rng_branches():
    """The random branches (torch.randperm / torch.multinomial / random.choice, target_mask.py:50-93) consume the
    generators exactly like the reference: same seeds, same masks."""
    from aline_amd.utils import create_target_mask
    cases = json.load(open(os.path.join(GOLDEN, "masks_rng.json")))["cases"]
    assert len(cases) >= 40
    for c in cases:
        torch.manual_seed(c["seed"]); np.random.seed(c["seed"]); random.seed(c["seed"])
        got = create_target_mask(**c["kwargs"]).tolist()
        assert got == c["mask"], (c["name"], c["seed"], got, c["mask"])
    # the draws do differ between seeds (the fixture is not degenerate)
    by = {}
    for c in cases:
        by.setdefault(c["name"], set()).add(tuple(c["mask"]))
    assert len(by["partial_theta"]) > 1 and len(by["predef_uniform"]) > 1 and len(by["split_random"]) > 1


def test_product_gp_kernels_and_psychometric(golden):
    """GPTask.kernel_matrix (the batched restatement of tasks/gaussian_process.py:194-342) and
    PsychometricTask.psychometric_function (tasks/psychometric.py:107-134) on the reference's values."""
    from aline_amd.tasks import GPTask, PsychometricTask
    fx = golden("tasks_r2")
    gp = GPTask(dim_x=2, embedding_type="mix", n_context_init=1, n_query_init=5, n_target_theta=3, n_target_data=4,
                device=torch.device("cpu"))
    x, ls, sc = fx.t("gp_x"), fx.t("gp_lengthscales"), fx.t("gp_scale")
    xb = x.unsqueeze(0).expand(4, -1, -1).contiguous()
    K = gp.kernel_matrix(xb, ls.unsqueeze(0).expand(4, -1), sc.reshape(1).expand(4), torch.arange(4))
    for i, kt in enumerate(GPTask.KERNELS):
        assert torch.allclose(K[i], fx.t("gp_K_" + kt), rtol=1e-5, atol=1e-6), kt
    ps = PsychometricTask(n_context_init=1, n_query_init=8, device=torch.device("cpu"))
    p = ps.psychometric_function(fx.t("psy_x"), fx.t("psy_theta"))
    assert torch.allclose(p, fx.t("psy_p"), rtol=1e-6, atol=1e-7)


def test_product_gmm_variance(golden):
    from aline_amd.utils import calculate_gmm_variance
    fx = golden("tasks_r2")
    m, s, w = fx.t("gmm_means"), fx.t("gmm_stds"), fx.t("gmm_weights")
    assert torch.allclose(calculate_gmm_variance(m, s, w), fx.t("gmm_var"), rtol=1e-6, atol=1e-6)
    assert torch.allclose(calculate_gmm_variance(m, s, w[:, 0]), fx.t("gmm_var_shared_w"), rtol=1e-6, atol=1e-6)

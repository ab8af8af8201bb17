"""Wide path (d_model = 256, 8 heads, bf16 MFMA): the fused per-step kernel (wide_step.h) against the generic
bf16 pipeline, the streamed block kernels and the fp32 pipeline -- same weights, same forced designs.

Tolerances are bf16 tolerances and are written where they are used: the three bf16 implementations round at
different places, so they agree with each other to a few 1e-2 in log-likelihood on O(1..10) values, and each
agrees with fp32 to the bound the committed reference fixture is checked against in test_hip_parity
(0.6 on max |d NLL| for plain bf16)."""
import os

import pytest
import torch

from helpers import native_model

pytestmark = pytest.mark.gpu

DIMS = {"dim_x": 2, "dim_y": 1, "d": 256, "F": 1024, "n_head": 8, "L": 2, "C": 10, "n_theta": 2,
        "embedding_type": "theta", "time_token": False}


def _run(prec, env, B, n_query, T, seed=5, select="forced", target_mask=None, dims=None, full=False):
    from aline_amd.rollout import Rollout
    from aline_amd.tasks import HiddenLocation
    from aline_amd import _lib
    with _lib.debug_env(env):
        model, _ = native_model(dims or DIMS, 11, prec)
        torch.manual_seed(seed)
        task = HiddenLocation(device=torch.device("cuda"), n_query_init=n_query)
        batch = task.sample_batch(B)
        if target_mask is not None:
            batch.target_mask = torch.tensor(target_mask, dtype=torch.bool, device="cuda")
        g = torch.Generator(device="cpu").manual_seed(seed)
        forced = torch.stack([torch.stack([torch.randint(0, n_query - t, (1,), generator=g)[0] for t in range(T)])
                              for _ in range(B)]).to("cuda")     # index into the queries remaining at step t
        ro = Rollout(model, batch, T, select=select, forced_idx=forced if select == "forced" else None).run()
        torch.cuda.synchronize()
        if full:
            return {k: getattr(ro, k).float().cpu().clone() for k in ("target_ll", "log_prob", "post_mean", "post_std", "post_weight")}
        return ro.target_ll.float().cpu().clone(), ro.log_prob.float().cpu().clone(), ro.idx.cpu().clone()


@pytest.mark.parametrize("B,n_query,T", [(3, 200, 6), (5, 37, 4), (2, 250, 3), (4, 16, 5)])
def test_step_kernel_matches_other_bf16_paths(B, n_query, T):
    """N = 1 + n_query + 2 tokens: 203 (13 tiles, the headline shape), 40 (partial tile), 253 (all 16 tiles),
    19 (two tiles, one wave idle)."""
    ll_s, lp_s, _ = _run("bf16", {}, B, n_query, T)
    ll_g, lp_g, _ = _run("bf16", {"ALINE_DISABLE_WIDE": "1"}, B, n_query, T)
    ll_f, lp_f, _ = _run("f32", {}, B, n_query, T)
    assert torch.isfinite(ll_s).all() and torch.isfinite(lp_s).all()
    # against the generic bf16 GEMM pipeline and against fp32: bf16 rounding noise only
    assert (ll_s - ll_g).abs().max() < 0.6 and (lp_s - lp_g).abs().max() < 0.1
    assert (ll_s - ll_f).abs().max() < 0.6 and (lp_s - lp_f).abs().max() < 0.1
    # no worse than the generic bf16 pipeline by more than its own error
    assert (ll_s - ll_f).abs().max() < 2.0 * (ll_g - ll_f).abs().max() + 0.05


def test_step_kernel_sampling_and_bookkeeping():
    """Sampled designs: finite log-probabilities, indices valid for the query set remaining at each step
    (base_task.py:133-154 removes the chosen query, so step t has n_query - t candidates)."""
    _, lp, idx = _run("bf16", {}, 6, 60, 8, select="sample")
    assert lp.shape == (6, 8) and (lp <= 0).all() and torch.isfinite(lp).all()
    for t in range(8):
        assert int(idx[:, t].min()) >= 0 and int(idx[:, t].max()) < 60 - t


def test_too_many_keys_falls_back_to_generic():
    """More than 64 visible keys (context + targets) is outside the wide kernels' register layout: the rollout
    must still run (generic bf16 pipeline) and agree with fp32 to bf16 tolerance."""
    ll_w, lp_w, _ = _run("bf16", {}, 2, 90, 70)
    ll_f, lp_f, _ = _run("f32", {}, 2, 90, 70)
    assert torch.isfinite(ll_w).all()
    assert (ll_w - ll_f).abs().max() < 1.5 and (lp_w - lp_f).abs().max() < 0.2


@pytest.mark.parametrize("mask", [[True, False], [False, True], [False, False]])
def test_step_kernel_with_target_mask(mask):
    """Queries attend only the selected targets (encoder.py:110-121); the key list of the fused kernel then has
    n_ctx + (#selected) entries.  An all-False mask leaves the context keys only."""
    ll_s, lp_s, _ = _run("bf16", {}, 3, 70, 5, target_mask=mask)
    ll_g, lp_g, _ = _run("bf16", {"ALINE_DISABLE_WIDE": "1"}, 3, 70, 5, target_mask=mask)
    ll_f, lp_f, _ = _run("f32", {}, 3, 70, 5, target_mask=mask)
    assert (ll_s - ll_g).abs().max() < 0.6 and (lp_s - lp_g).abs().max() < 0.1
    assert (ll_s - ll_f).abs().max() < 0.6 and (lp_s - lp_f).abs().max() < 0.1
    # the mask must matter: a different selection changes the query logits
    _, lp_other, _ = _run("bf16", {}, 3, 70, 5, target_mask=[not m for m in mask] if any(mask) else [True, True])
    assert (lp_s - lp_other).abs().max() > 1e-4


def test_wide_dims_outside_the_step_kernel_use_block_kernels():
    """F = 2048: the per-layer parameters no longer fit the step kernel's LDS budget -> streamed block kernels
    (same tile images, activations in HBM); still bf16-close to fp32."""
    global DIMS
    saved = dict(DIMS)
    try:
        DIMS = dict(DIMS, F=2048, L=1)
        ll_w, lp_w, _ = _run("bf16", {}, 2, 40, 3)
        ll_f, lp_f, _ = _run("f32", {}, 2, 40, 3)
    finally:
        DIMS = saved
    assert torch.isfinite(ll_w).all()
    assert (ll_w - ll_f).abs().max() < 0.6 and (lp_w - lp_f).abs().max() < 0.1


def test_step_kernel_is_reproducible():
    """Same inputs, same bits (the kernel has barriers between every LDS producer and consumer; a missing one
    showed up as run-to-run differences during development)."""
    a = _run("bf16", {}, 3, 200, 6)
    b = _run("bf16", {}, 3, 200, 6)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_block_kernels_cross_check():
    """ALINE_WIDE_BLOCKS=1 forces the streamed per-block kernels (the fallback for N > 256 or large F): same operand
    rounding and accumulation order as the fused step kernel almost everywhere (the FFN bias is added before /
    after the fp32 accumulation, which flips an occasional bf16 rounding), and reproducible."""
    ll_s, lp_s, _ = _run("bf16", {}, 3, 200, 6)
    ll_b, lp_b, _ = _run("bf16", {"ALINE_WIDE_BLOCKS": "1"}, 3, 200, 6)
    ll_b2, lp_b2, _ = _run("bf16", {"ALINE_WIDE_BLOCKS": "1"}, 3, 200, 6)
    assert (ll_s - ll_b).abs().max() < 0.1 and (lp_s - lp_b).abs().max() < 5e-2
    assert torch.equal(ll_b, ll_b2) and torch.equal(lp_b, lp_b2)


@pytest.mark.parametrize("C", [10, 16])
def test_wide_posterior_rows_are_the_right_rows(C):
    """Structure-sensitive check of the GMM stage (the absolute bf16 bounds above would let an indexing slip through):
    per component, the posterior parameters of the wide path follow those of the fp32 pipeline row by row -- a
    misplaced row or component (e.g. a raw-output stride too small for C = 16: 48 floats per row) destroys the
    correlation, bf16 rounding does not -- and the mixture weights of every row still sum to one."""
    dims = dict(DIMS, C=C)
    w = _run("bf16", {}, 3, 60, 4, dims=dims, full=True)
    f = _run("f32", {}, 3, 60, 4, dims=dims, full=True)
    assert w["post_mean"].shape[-1] == C
    assert (w["post_weight"].sum(-1) - 1).abs().max() < 1e-5
    assert (w["post_std"] > 0).all()
    for k in ("post_mean", "post_std", "post_weight"):
        a, b = w[k].reshape(-1, C), f[k].reshape(-1, C)
        for c in range(C):
            corr = torch.corrcoef(torch.stack([a[:, c], b[:, c]]))[0, 1]
            assert corr > 0.98, (k, c, float(corr))
        rel = (a - b).abs().median() / b.abs().median()
        assert rel < 0.05, (k, float(rel))

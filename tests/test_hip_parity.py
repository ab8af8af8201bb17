"""GPU parity tests proper: the HIP path (through the C ABI, via the drop-in nn.Modules) against
(1) the golden vectors the reference itself produced and (2) the CPU oracle, on the same inputs,
teacher-forced with the reference's own design sequence.

Tolerances (fp32 unless noted):
  posterior NLL            |delta| <= 1e-4   (north-star tolerance, BASELINE.json)
  zt / GMM params          atol 5e-5 (d=32), 2e-4 (d>=256: longer fp32 reductions)
  log_prob                 atol 1e-4
  bf16x3 (split-bf16 MFMA): NLL 1e-3;  bf16 single pass: NLL 0.6 (stated bounds, not claimed as parity)
"""
import pytest
import torch

import aline_oracle as orc
from conftest import MODEL_FIXTURES
from helpers import maxdiff, native_model, to_dev

pytestmark = pytest.mark.gpu

NLL_TOL = 1e-4


def tols(dims):
    # zt / GMM parameter tolerance.  CES designs live in [0, 100]^6 (tasks/ces.py:86-92): embeddings
    # and attention scores are ~1e2..1e4 in magnitude, so fp32 rounding alone moves near-tied softmax
    # weights by ~1e-3; the NLL bound (1e-4) is unchanged for every fixture.
    if dims["dim_x"] == 6:
        return dict(p=1e-3)
    return dict(p=2e-4 if dims["d"] >= 256 else 5e-5)


@pytest.mark.parametrize("name", MODEL_FIXTURES)
def test_embed_encode_stages(golden, name):
    """Embedder.forward / Encoder.forward stand-alone entry points vs the reference tensors."""
    fx = golden(name)
    model, _ = native_model(fx.meta["dims"], fx.meta["wseed"])
    batch = to_dev(fx.batch())
    with torch.no_grad():
        emb = model.embedder(batch)
        ref = fx.t("embedding_0")
        assert torch.allclose(emb.cpu(), ref, rtol=2e-6, atol=3e-5), maxdiff(emb, ref)
        z = model.encoder(batch, emb)
    tol = 2e-4 if fx.meta["dims"]["d"] >= 256 else 5e-5
    assert maxdiff(z, fx.t("encoding_eval_0")) < tol
    assert maxdiff(z, fx.t("encoding_train_0")) < tol


@pytest.mark.parametrize("mode", ["train", "eval"])
@pytest.mark.parametrize("name", MODEL_FIXTURES)
def test_step_api_teacher_forced(golden, name, mode):
    """model.forward(batch) + Task.update_batch, step by step, against the reference outputs."""
    from aline_amd.tasks import Task
    from aline_amd.utils import compute_ll
    fx = golden(name)
    dims, T = fx.meta["dims"], fx.meta["T"]
    model, _ = native_model(dims, fx.meta["wseed"])
    model.train(mode == "train")
    task = Task(dim_x=dims["dim_x"], dim_y=1)
    batch = to_dev(fx.batch())
    forced = fx.forced_idx(mode).cuda()
    tp = tols(dims)["p"]
    with torch.no_grad():
        for t in range(T):
            if dims.get("time_token"):
                batch.t = torch.tensor([t / T], device="cuda")
            out = model.forward(batch, forced_idx=forced[:, t])
            d, p = out.design_out, out.posterior_out
            assert maxdiff(d.zt, fx.t(f"{mode}.zt_{t}")) < tp
            assert abs(float(d.zt.sum(-1).mean()) - 1.0) < 1e-5
            assert (d.idx.cpu() == fx.t(f"{mode}.idx_{t}")).all()
            assert maxdiff(d.log_prob, fx.t(f"{mode}.log_probs")[:, t]) < 1e-4
            assert maxdiff(p.mixture_means, fx.t(f"{mode}.means_{t}")) < tp
            assert maxdiff(p.mixture_stds, fx.t(f"{mode}.stds_{t}")) < tp
            assert maxdiff(p.mixture_weights, fx.t(f"{mode}.weights_{t}")) < tp
            batch = task.update_batch(batch, d.idx)
            ll = compute_ll(batch.target_all, p.mixture_means, p.mixture_stds, p.mixture_weights)
            assert maxdiff(ll, fx.t(f"{mode}.target_ll_{t}")) < NLL_TOL
            if t in (0, T - 1) and f"{mode}.pq_means_{t}" in fx:
                pq = out.posterior_out_query          # lazily computed on access
                assert maxdiff(pq.mixture_means, fx.t(f"{mode}.pq_means_{t}")) < tp
                assert maxdiff(pq.mixture_stds, fx.t(f"{mode}.pq_stds_{t}")) < tp
                assert maxdiff(pq.mixture_weights, fx.t(f"{mode}.pq_weights_{t}")) < tp
    assert maxdiff(batch.context_x, fx.t(f"{mode}.final_context_x")) == 0.0
    assert maxdiff(batch.context_y, fx.t(f"{mode}.final_context_y")) == 0.0


# Measured on MI355X over all fixtures: f32 <= 3e-5, bf16x3 1.1e-4..4.0e-4, bf16 0.06..0.40.  Only
# f32 meets the north-star 1e-4 NLL bound; the bf16 modes are throughput modes with stated bounds.
# f16x3 (3-term f16 split, x3.h / gemm.h PREC 3) is the second reference-precision mode: same 1e-4 bound as f32.
@pytest.mark.parametrize("precision,nll_tol", [("f32", 1e-4), ("f16x3", 1e-4), ("bf16x3", 1e-3), ("bf16", 0.6)])
@pytest.mark.parametrize("name", MODEL_FIXTURES)
def test_rollout_api_teacher_forced(golden, name, precision, nll_tol):
    """Static-slot rollout (one C call for T steps) vs the reference: NLLs, log-probs, designs."""
    from aline_amd.rollout import Rollout
    fx = golden(name)
    dims, T = fx.meta["dims"], fx.meta["T"]
    model, _ = native_model(dims, fx.meta["wseed"], precision)
    batch = to_dev(fx.batch())
    mode = "train"
    ro = Rollout(model, batch, T, select="forced", forced_idx=fx.forced_idx(mode),
                 time_token_T=T if dims.get("time_token") else 0, keep_zt=True).run()
    torch.cuda.synchronize()
    ref_ll = torch.stack([fx.t(f"{mode}.target_ll_{t}") for t in range(T)])      # [T, B, n_t]
    assert maxdiff(ro.target_ll, ref_ll) < nll_tol
    nll_q, nll = ro.nlls(dims["embedding_type"], fx.meta["mask_type"])
    assert maxdiff(nll, fx.t(f"{mode}.nll")) < nll_tol
    assert maxdiff(nll_q, fx.t(f"{mode}.nll_q")) < nll_tol
    assert maxdiff(ro.log_prob, fx.t(f"{mode}.log_probs")) < {"f32": 2e-4, "f16x3": 2e-4, "bf16x3": 2e-3, "bf16": 1.5}[precision]
    assert (ro.idx.cpu() == fx.forced_idx(mode)).all()
    if precision in ("f32", "f16x3"):
        for t in range(T):
            ref = fx.t(f"{mode}.zt_{t}")
            assert maxdiff(ro.zt[t, :, :ref.shape[1]], ref) < tols(dims)["p"]
            assert float(ro.zt[t, :, ref.shape[1]:].abs().max()) == 0.0 if ref.shape[1] < ro.zt.shape[2] else True
    cx, cy = ro.export_context()
    assert maxdiff(cx, fx.t(f"{mode}.final_context_x")) == 0.0
    assert maxdiff(cy, fx.t(f"{mode}.final_context_y")) == 0.0


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
@pytest.mark.parametrize("name", MODEL_FIXTURES)
def test_rollout_query_posterior_matches_reference(golden, name, precision):
    """posterior_out_query (model/head.py:366) of the first and the last step from the rollout API (`postq_*`, by slot)
    against what the reference's forward returned at those steps (fixture `pq_*`, rows = the remaining queries in
    their order-preserving compaction, tasks/base_task.py:114-117)."""
    from aline_amd.rollout import Rollout
    fx = golden(name)
    dims, T = fx.meta["dims"], fx.meta["T"]
    model, _ = native_model(dims, fx.meta["wseed"], precision)
    ro = Rollout(model, to_dev(fx.batch()), T, select="forced", forced_idx=fx.forced_idx("train"),
                 time_token_T=T if dims.get("time_token") else 0, keep_query_posterior=True).run()
    torch.cuda.synchronize()
    # (round 4: the x3 / x5 tile-image paths serve the request too; only the exact-fp32 fused kernel hands it to the generic pipeline)
    assert ro.path in ("s3::step_kernel", "x3::layer_kernel", "x5::layer_kernel", "generic pipeline")
    if precision == "f16x3" and dims["d"] in (256, 512):
        assert ro.path == {256: "x3::layer_kernel", 512: "x5::layer_kernel"}[dims["d"]]
    role = ro.role.cpu()
    tp = tols(dims)["p"]
    for t in (0, T - 1):
        cand = ~((role > 0) & (role <= ro.n_c0 + t))                  # slot is still a query at step t
        for b in range(ro.B):
            sl = torch.where(cand[b])[0]
            for got, key in ((ro.postq_mean, "pq_means"), (ro.postq_std, "pq_stds"), (ro.postq_weight, "pq_weights")):
                ref = fx.t(f"train.{key}_{t}")[b]
                assert ref.shape[0] == len(sl)
                assert maxdiff(got[t, b, sl], ref) < 2 * tp, (key, t, b)


@pytest.mark.parametrize("name", ["cfg2_location_d32", "cfg1_almix_d1_data", "cfg4_ces"])
def test_rollout_argmax_matches_reference_designs(golden, name):
    """Free-running eval rollout: design sequence agrees with the reference's argmax trajectory."""
    from aline_amd.rollout import Rollout
    fx = golden(name)
    model, _ = native_model(fx.meta["dims"], fx.meta["wseed"])
    ro = Rollout(model, to_dev(fx.batch()), fx.meta["T"], select="argmax").run()
    agree = (ro.idx.cpu() == fx.forced_idx("eval")).float().mean()
    assert agree >= 0.95, float(agree)


def test_rollout_sampling_statistics(golden):
    """SAMPLE mode: inverse-CDF draws follow zt (chi-square-free check: empirical frequency of the
    most likely design over many uniform draws), log_prob = log zt[idx]."""
    from aline_amd.rollout import Rollout
    fx = golden("cfg2_location_d32")
    model, _ = native_model(fx.meta["dims"], fx.meta["wseed"])
    model.train()
    b1 = {k: v[:1].repeat(4096, 1, 1) if v.dim() == 3 else v for k, v in fx.batch().items()}
    ro = Rollout(model, to_dev(b1), 1, select="sample", keep_zt=True).run()
    zt = ro.zt[0, 0].cpu()
    idx = ro.idx[:, 0].cpu()
    freq = torch.bincount(idx, minlength=zt.numel()).float() / idx.numel()
    assert float((freq - zt).abs().max()) < 0.02
    lp = ro.log_prob[:, 0].cpu()
    assert torch.allclose(lp, torch.log(zt[idx]), atol=1e-5)


def test_oracle_cross_check_random_inputs():
    """Fresh seeded inputs (not in the fixtures): HIP path vs the CPU oracle, ragged sizes."""
    from aline_amd.rollout import Rollout
    dims = dict(dim_x=3, dim_y=1, d=64, F=96, n_head=4, L=2, C=7, n_theta=3, embedding_type="mix",
                time_token=False)
    model, sd = native_model(dims, 99)
    g = torch.Generator().manual_seed(5)
    B, n_c, n_q, n_td, T = 5, 3, 37, 11, 6
    batch = dict(context_x=torch.randn(B, n_c, 3, generator=g), context_y=torch.randn(B, n_c, 1, generator=g),
                 query_x=torch.randn(B, n_q, 3, generator=g), query_y=torch.randn(B, n_q, 1, generator=g),
                 target_x=torch.randn(B, n_td, 3, generator=g),
                 target_all=torch.randn(B, n_td + 3, 1, generator=g),
                 target_mask=torch.rand(n_td + 3, generator=g) > 0.5)
    cfg = dict(embedding_type="mix", n_head=4, num_layers=2, num_components=7, std_min=1e-4,
               n_target_theta=3)
    forced = torch.stack([torch.randint(0, n_q - t, (B,), generator=g) for t in range(T)], 1)
    ref = orc.rollout(sd, batch, cfg, T, forced_idx=forced, mask_type="partial")
    ro = Rollout(model, to_dev(batch), T, select="forced", forced_idx=forced, keep_zt=True).run()
    assert maxdiff(ro.target_ll, torch.stack(ref["target_ll"])) < NLL_TOL
    assert maxdiff(ro.log_prob, torch.stack(ref["log_prob"], 1)) < 1e-4


def test_no_cpu_fallback():
    from aline_amd import Aline, Embedder, Encoder, OutputHead
    from aline_amd.utils import AttrDict
    model = Aline(Embedder(2, 1, 32, 128, 2, "theta"), Encoder(32, 128, 4, 0.0, 3),
                  OutputHead(2, 1, 32, 128)).cuda()
    cpu_batch = AttrDict(context_x=torch.zeros(2, 1, 2), context_y=torch.zeros(2, 1, 1),
                         query_x=torch.zeros(2, 5, 2), query_y=torch.zeros(2, 5, 1),
                         target_all=torch.zeros(2, 2, 1))
    with torch.no_grad(), pytest.raises(RuntimeError):
        model.forward(cpu_batch)


def test_layer_tail_and_gmm_kernels_match_the_per_op_pipeline(golden):
    """d=32 / F=128 generic pipeline: `fused::layer_tail_kernel` (out-proj + LN1 + FFN + LN2 in one kernel) and the
    row-mapped `fused::gmm_rows_kernel` against the per-op kernels they replace (ALINE_DBG_NO_LAYER_TAIL), mix mode
    with a split mask, teacher-forced with the same designs."""
    import os
    from aline_amd.rollout import Rollout
    fx = golden("cfg3_almix_d2")
    dims, T = fx.meta["dims"], fx.meta["T"]
    model, _ = native_model(dims, fx.meta["wseed"])
    batch, forced = to_dev(fx.batch()), fx.forced_idx("train")
    outs = []
    from aline_amd import _lib
    for flags in ([], ["NO_LAYER_TAIL"]):
        with _lib.debug(*flags):
            ro = Rollout(model, batch, T, select="forced", forced_idx=forced, keep_zt=True).run()
            torch.cuda.synchronize()
        outs.append((ro.target_ll.cpu().clone(), ro.log_prob.cpu().clone(), ro.post_std.cpu().clone()))
    assert float((outs[0][0] - outs[1][0]).abs().max()) < 1e-4
    assert float((outs[0][1] - outs[1][1]).abs().max()) < 1e-4
    assert torch.allclose(outs[0][2], outs[1][2], rtol=1e-4, atol=1e-6)

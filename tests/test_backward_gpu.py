"""GPU: gradients of the training objective (train_aline.py:113-132) from the native backward vs the
reference's autograd gradients stored in the golden fixtures (loss = design_loss + predict_loss,
alpha = gamma = 1, teacher-forced designs)."""
import pytest
import torch

from helpers import grad_errors, native_model, to_dev

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["cfg2_location_d32", "cfg1_almix_d1_data", "aux_timetoken_train",
                                  "grad_cfg2_d256", "grad_cfg5_d512", "grad_cfg3_split", "grad_cfg4_ces"])
def test_gradients_match_reference_autograd(golden, name):
    """(aux_timetoken_train: data mode with the time token of model/head.py:342-345 -- the acquisition MLP sees [z | t / T],
    train_aline.py:80-82 -- incl. the gradient of the time column of its first layer; fixture of oracle/make_golden_r3.py.
    grad_*: round-4 fixtures of oracle/make_golden_r4.py -- d = 256 / F = 1024 / head_dim 32, d = 512 / head_dim 64 with the
    predefined mask, the cfg3 split mask (103 targets, 304 rows) and CES (dim_x = 6): the widths and masks the training numbers of
    the bench line and of tools/config_bench.py are quoted on.)"""
    from aline_amd.train import train_step
    fx = golden(name)
    dims, T = fx.meta["dims"], fx.meta["T"]
    model, _ = native_model(dims, fx.meta["wseed"])
    terms, ro = train_step(model, to_dev(fx.batch()), T, optimizer=None, embedding_type=dims["embedding_type"],
                           mask_type=fx.meta["mask_type"], forced_idx=fx.forced_idx("train"), clip_grads=False)
    torch.cuda.synchronize()
    assert abs(float(terms["predict_loss"]) - float(fx.np("train.predict_loss"))) < 1e-4
    assert abs(float(terms["design_loss"]) - float(fx.np("train.design_loss"))) < 5e-3
    # the batch-normalised rewards themselves (train_aline.py:113-122): z-scores of clamped NLL gains over B = 2 ... 8
    # episodes, so an NLL difference of 1e-5 next to the clamp moves an entry by ~1e-3
    assert float((terms["R"].cpu() - fx.t("train.R")).abs().max()) < 5e-3
    named = [(k, p.grad) for k, p in model.named_parameters()]
    if "train64.design_loss" in fx:
        # round-4 fixtures carry the reference's gradients twice: its fp32 run and the same rollout in fp64.  The reference's own
        # fp32 rounding is up to 1.2e-3 of a parameter's max |grad| there (CES, d = 256 acquisition head: measured against its
        # fp64 run), so the fp64 gradients are the 1e-3 target and the fp32 ones are held at 3e-3.
        worst64 = max(grad_errors(fx, named, "train64").items(), key=lambda kv: kv[1])
        worst32 = max(grad_errors(fx, named, "train").items(), key=lambda kv: kv[1])
        if worst64[1] < 1e-3 and worst32[1] < 3e-3:
            return
        # Knife-edge ReLU gates: ~1e6 gates per rollout, so some pre-activation always sits within fp32 rounding of zero (the
        # fp64 oracle finds |h| ~ 1e-6 of the layer's mean |h| in every one of these fixtures), an fp32 forward may take such a
        # gate either way, and ONE flipped gate moves its hidden unit's weight row by ~1e-3 of the tensor's max |grad|.  The
        # arbiter is the same as in test_fused_backward_kernels_in_mix_mode_with_target_data_keys: the fp64 oracle
        # differentiated under the flips of its knife-edge gates spans what a correct kernel may return.
        worst = _fixture_residual_after_gate_flips(fx, model, terms, ro, named)
        assert worst[1] < 1e-3, (worst, worst64, worst32)
        return
    worst = max(grad_errors(fx, named).items(), key=lambda kv: kv[1])
    # measured on MI355X: <= 4e-5 of each parameter's max |grad| (the acquisition output bias has a
    # mathematically zero gradient -- softmax shift invariance -- hence the absolute floor in `scale`)
    assert worst[1] < 1e-3, worst


def _fixture_residual_after_gate_flips(fx, model, terms, ro, named, eps=5e-6):
    """Per-parameter error against the fixture's fp64 reference gradients at the positions the fixture holds (whole small tensors,
    the seeded sample of large ones), after removing the best 0 / 1 combination of the oracle's knife-edge gate flips."""
    sd = {k: v.detach().cpu().double().requires_grad_(True) for k, v in model.state_dict().items()}
    cpu = {k: (v.double() if v.is_floating_point() else v) for k, v in fx.batch().items()}
    g0, deltas, ref, n_gates = _oracle_gradients_with_gate_flips(sd, cpu, fx.cfg(), fx.meta["T"], fx.forced_idx("train"),
                                                                 fx.meta["mask_type"], terms["g_logp"].cpu().double(),
                                                                 terms["g_ll"].cpu().double(), eps=eps)
    assert len(deltas) <= 64 and len(deltas) < 1e-4 * n_gates, (len(deltas), n_gates)      # a handful of gates out of millions
    got = dict(named)
    pos, refv, gotv, spans, off = [], [], [], [], 0
    for k, v in sd.items():
        n = v.numel()
        if "train64.grad." + k in fx:
            idx, val = torch.arange(n), fx.t("train64.grad." + k).double().reshape(-1)
        else:
            idx, val = fx.t("train64.gidx." + k), fx.t("train64.gval." + k).double()
        pos.append(off + idx)
        refv.append(val)
        gotv.append(got[k].detach().cpu().double().reshape(-1)[idx])
        spans.append((k, len(idx), max(float(fx.np("train64.gmax." + k)), 1e-4)))
        off += n
    pos = torch.cat(pos)
    res, flips = _explain_with_gate_flips(torch.cat(gotv) - torch.cat(refv), [dl[pos] for dl in deltas])
    worst, o = ("", 0.0), 0
    for k, n, scale in spans:
        err = float(res[o:o + n].abs().max()) / scale
        if err > worst[1]:
            worst = (k, err)
        o += n
    return worst


def test_backward_chunking_is_consistent(golden):
    """t_chunk only changes the batching of the (independent) steps, not the result."""
    from aline_amd.train import train_step
    fx = golden("cfg2_location_d32")
    dims, T = fx.meta["dims"], 6
    forced = fx.forced_idx("train")[:, :T]
    grads = []
    for tc in (T, 2):
        model, _ = native_model(dims, fx.meta["wseed"])
        train_step(model, to_dev(fx.batch()), T, forced_idx=forced, clip_grads=False, t_chunk=tc)
        grads.append(torch.cat([p.grad.reshape(-1) for p in model.parameters()]).cpu())
    assert float((grads[0] - grads[1]).abs().max()) <= 1e-5 * float(grads[0].abs().max()) + 1e-7


def test_reference_style_training_loop_with_autograd(golden):
    """The reference's own loop body (train_aline.py:80-132) written against the drop-in modules:
    model.forward(batch) per step, Task.update_batch, a *torch* compute_ll (utils/eval.py:200-207 is plain
    differentiable torch code), loss.backward().  Gradients must equal the reference's."""
    from aline_amd.tasks import Task
    from aline_amd.utils import select_targets_by_mask
    fx = golden("cfg2_location_d32")
    dims, T = fx.meta["dims"], fx.meta["T"]
    model, _ = native_model(dims, fx.meta["wseed"])
    model.train()
    task = Task(dim_x=dims["dim_x"], dim_y=1)
    batch = to_dev(fx.batch())
    forced = fx.forced_idx("train").cuda()

    def compute_ll(value, means, stds, weights):          # reference utils/eval.py:200-207
        comp = torch.distributions.Normal(means, stds, validate_args=False)
        return torch.logsumexp(comp.log_prob(value) + torch.log(weights), dim=-1)

    log_probs, nlls, nlls_q = [], [], []
    for t in range(T):
        pred = model.forward(batch, forced_idx=forced[:, t])
        d, p = pred.design_out, pred.posterior_out
        batch = task.update_batch(batch, d.idx)
        log_probs.append(d.log_prob)
        ll = compute_ll(batch.target_all, p.mixture_means, p.mixture_stds, p.mixture_weights)
        masked = select_targets_by_mask(ll, batch.target_mask)
        nlls_q.append(-masked.mean(dim=-1))
        nlls.append(-ll.mean(dim=-1))
    log_probs = torch.stack(log_probs, 1)
    R = torch.stack([torch.clamp(nlls_q[t - 1] - nlls_q[t], min=0.0).detach() for t in range(1, T)], 1)
    R = (R - R.mean(dim=0, keepdim=True)) / (R.std(dim=0, keepdim=True) + 1e-9)
    design_loss = -torch.mean(log_probs[:, :-1] * R)
    predict_loss = torch.mean(torch.stack(nlls))
    (design_loss + predict_loss).backward()
    torch.cuda.synchronize()
    assert abs(float(predict_loss) - float(fx.np("train.predict_loss"))) < 1e-4
    worst = ("", 0.0)
    for k, prm in model.named_parameters():
        ref = fx.t("train.grad." + k)
        err = float((prm.grad.cpu() - ref).abs().max()) / max(float(ref.abs().max()), 1e-4)
        if err > worst[1]:
            worst = (k, err)
    assert worst[1] < 1e-3, worst


class _RefComposition(torch.nn.Module):
    """The reference's own composition, model/base.py:32-50, verbatim in structure: three hydra-instantiated modules and
    `head(batch, encoder(batch, embedder(batch)))` -- what `train_aline.py:246-249` builds when only the `_target_` strings
    of config/{embedder,encoder,head}/*.yaml are switched to the aline_amd classes."""

    def __init__(self, embedder, encoder, head):
        super().__init__()
        self.embedder, self.encoder, self.head = embedder, encoder, head

    def forward(self, batch):
        return self.head(batch, self.encoder(batch, self.embedder(batch)))


@pytest.mark.parametrize("name", ["cfg2_location_d32", "cfg1_almix_d1_data"])
def test_target_only_swap_trains_under_reference_composition(golden, name):
    """Stage backward entry points (aline_embed_backward / aline_encoder_backward / aline_head_backward): the stand-alone
    Embedder, Encoder and OutputHead composed by the reference's own Aline class, the reference's training-loop body
    (train_aline.py:80-132) and `loss.backward()` -- gradients against the reference's autograd gradients."""
    from aline_amd import Embedder, Encoder, OutputHead
    from aline_amd.tasks import Task
    from aline_amd.utils import select_targets_by_mask
    import aline_oracle as orc
    fx = golden(name)
    dims, T = fx.meta["dims"], fx.meta["T"]
    model = _RefComposition(
        Embedder(dims["dim_x"], dims["dim_y"], dims["d"], dims["F"], dims["n_theta"], dims["embedding_type"]),
        Encoder(dims["d"], dims["F"], dims["n_head"], 0.0, dims["L"]),
        OutputHead(dims["dim_x"], dims["dim_y"], dims["d"], dims["F"], num_components=dims["C"]))
    model.load_state_dict(orc.make_state_dict(fx.meta["wseed"], **dims), strict=True)
    model = model.cuda().train()
    task = Task(dim_x=dims["dim_x"], dim_y=1)
    batch = to_dev(fx.batch())
    forced = fx.forced_idx("train").cuda()

    def compute_ll(value, means, stds, weights):          # reference utils/eval.py:200-207
        comp = torch.distributions.Normal(means, stds, validate_args=False)
        return torch.logsumexp(comp.log_prob(value) + torch.log(weights), dim=-1)

    n_th = dims["n_theta"]
    log_probs, nlls, nlls_q = [], [], []
    for t in range(T):
        z = model.encoder(batch, model.embedder(batch))
        pred = model.head(batch, z, forced_idx=forced[:, t])      # (teacher forcing: the fixture's designs)
        d, p = pred.design_out, pred.posterior_out
        batch = task.update_batch(batch, d.idx)
        log_probs.append(d.log_prob)
        ll = compute_ll(batch.target_all, p.mixture_means, p.mixture_stds, p.mixture_weights)
        masked = select_targets_by_mask(ll, batch.target_mask)
        nlls_q.append(-masked.mean(dim=-1))
        if dims["embedding_type"] == "mix":
            nlls.append(-(ll[:, :-n_th].mean(-1) + ll[:, -n_th:].mean(-1)))
        else:
            nlls.append(-ll.mean(dim=-1))
    log_probs = torch.stack(log_probs, 1)
    R = torch.stack([torch.clamp(nlls_q[t - 1] - nlls_q[t], min=0.0).detach() for t in range(1, T)], 1)
    R = (R - R.mean(dim=0, keepdim=True)) / (R.std(dim=0, keepdim=True) + 1e-9)
    design_loss = -torch.mean(log_probs[:, :-1] * R)
    predict_loss = torch.mean(torch.stack(nlls))
    (design_loss + predict_loss).backward()
    torch.cuda.synchronize()
    assert abs(float(predict_loss) - float(fx.np("train.predict_loss"))) < 1e-4
    worst = ("", 0.0)
    for k, prm in model.named_parameters():
        ref = fx.t("train.grad." + k)
        assert prm.grad is not None, k
        err = float((prm.grad.cpu() - ref).abs().max()) / max(float(ref.abs().max()), 1e-4)
        if err > worst[1]:
            worst = (k, err)
    assert worst[1] < 1e-3, worst
    # and forward() of the composition itself is the reference's call (base.py:47-50)
    with torch.no_grad():
        out = model(to_dev(fx.batch()))
    assert out.design_out.zt.shape[0] == fx.meta["B"]


@pytest.mark.parametrize("T,B", [(30, 24), (44, 6), (60, 4)])
def test_fused_backward_kernels_match_per_op_pipeline(T, B):
    """The fused training-backward kernels of the small-width model (tail_bwd.h: token-local tail; attn_bwd_mfma.h:
    in-projection + attention, <= 32 keys at T = 30, <= 48 at T = 44, beyond that the per-op attention kernels;
    acq_head_bwd.h: acquisition head; layer_fwd.h: the forward recompute of a layer) against the per-op pipeline (GEMM / LayerNorm / attention kernels with saved
    activations), same rollout, same upstream gradients."""
    from aline_amd import Aline, Embedder, Encoder, OutputHead
    from aline_amd.rollout import Rollout
    from aline_amd.tasks import HiddenLocation
    from aline_amd.train import backward, reinforce_terms
    torch.manual_seed(5)
    model = Aline(Embedder(2, 1, 32, 128, 2, "theta"), Encoder(32, 128, 4, 0.0, 3), OutputHead(2, 1, 32, 128)).cuda()
    batch = HiddenLocation().sample_batch(B)
    with torch.no_grad():
        ro = Rollout(model, batch, T, select="sample").run()
        terms = reinforce_terms(ro, "theta")
        grads = []
        from aline_amd import _lib
        for flags in ([], ["NO_BWD_TAIL", "NO_BWD_ATTN_BLOCK", "NO_BWD_ACQ", "NO_BWD_LAYER_FWD", "NO_BWD_GMM128", "NO_BWD_GMM_BATCHED", "NO_BWD_GMM_FUSED"]):
            with _lib.debug(*flags):
                for p in model.parameters():
                    p.grad = None
                backward(model, ro, terms["g_logp"], terms["g_ll"])
                torch.cuda.synchronize()
            grads.append({k: p.grad.clone() for k, p in model.named_parameters()})
    worst = ("", 0.0)
    # (the acquisition output bias has a mathematically zero gradient -- softmax shift invariance: rounding noise on both sides)
    floor = 1e-2 * max(float(g.abs().max()) for g in grads[1].values())
    for k in grads[0]:
        ref = grads[1][k]
        err = float((grads[0][k] - ref).abs().max()) / max(float(ref.abs().max()), floor)
        if err > worst[1]:
            worst = (k, err)
    # both sides are exact-fp32 products in different summation orders
    assert worst[1] < 2e-4, worst


@pytest.mark.parametrize("d,F,H,B,T,nq", [(64, 256, 4, 70, 3, 20), (256, 1024, 8, 5, 2, 12)])
def test_wide_gmm_head_backward_matches_the_per_row_kernel(d, F, H, B, T, nq):
    """GMM head backward at F > 128 (the d = 256 / 512 models): `gmm_bwd_wide_kernel` (d ll / d raw per row into LDS, then weight-gradient
    partials per hidden unit in registers, one atomic per element and workgroup) against `gmm_bwd_kernel`'s per-row atomics it replaces
    (15 ms per call at d = 256 / F = 1024) -- every gradient of the model, same rollout, same upstream gradients; 70 episodes x 2 targets
    are more rows than one 64-row workgroup."""
    from aline_amd import Aline, Embedder, Encoder, OutputHead, _lib
    from aline_amd.rollout import Rollout
    from aline_amd.tasks import HiddenLocation
    from aline_amd.train import backward, reinforce_terms
    torch.manual_seed(9)
    model = Aline(Embedder(2, 1, d, F, 2, "theta"), Encoder(d, F, H, 0.0, 2), OutputHead(2, 1, d, F)).cuda()
    batch = HiddenLocation(n_query_init=nq).sample_batch(B)
    with torch.no_grad():
        ro = Rollout(model, batch, T, select="sample").run()
        terms = reinforce_terms(ro, "theta")
        grads = []
        for flags in ([], ["NO_BWD_GMM_WIDE"]):
            with _lib.debug(*flags):
                for p in model.parameters():
                    p.grad = None
                backward(model, ro, terms["g_logp"], terms["g_ll"])
                torch.cuda.synchronize()
            grads.append({k: p.grad.clone() for k, p in model.named_parameters()})
    worst = ("", 0.0)
    floor = 1e-2 * max(float(g.abs().max()) for g in grads[1].values())
    for k in grads[0]:
        ref = grads[1][k]
        assert torch.isfinite(grads[0][k]).all(), k
        err = float((grads[0][k] - ref).abs().max()) / max(float(ref.abs().max()), floor)
        if err > worst[1]:
            worst = (k, err)
    assert worst[1] < 2e-4, worst       # the same fp32 products in another summation order


@pytest.mark.parametrize("emb,T,B,kw", [("theta", 30, 24, {}), ("theta", 9, 7, {"t_chunk": 4}), ("mix", 12, 6, {"n_td": 100}), ("mix", 8, 5, {"n_td": 20, "mask": "split"})])
def test_backward_from_the_activations_the_s3_rollout_saved(emb, T, B, kw):
    """aline_rollout.saved_acts: the training rollout on the s3 path writes every layer's input and attention output (fp32 rows) and
    the backward reads them instead of recomputing the layers (layer_fwd.h / the per-op recompute).  Same rollout, same upstream
    gradients, against the recomputing backward (NO_BWD_SAVED_ACTS): the saved values are the f16x3 forward's (fp32-grade), the
    recomputed ones exact fp32 -- gradients agree to the reference-precision bound; with more than 48 keys (mix mode, 100 data
    targets) the per-op attention backward runs on the saved rows too; a chunked backward (t_chunk < T) reads its slice."""
    from aline_amd import Aline, Embedder, Encoder, OutputHead, _lib
    from aline_amd.rollout import Rollout
    from aline_amd.tasks import GPTask, HiddenLocation
    from aline_amd.train import backward, reinforce_terms
    from aline_amd.utils import create_target_mask
    torch.manual_seed(12)
    if emb == "theta":
        model = Aline(Embedder(2, 1, 32, 128, 2, "theta"), Encoder(32, 128, 4, 0.0, 3), OutputHead(2, 1, 32, 128)).cuda()
        batch = HiddenLocation(n_query_init=60).sample_batch(B)
        mask_type = "all"
    else:
        n_td = kw["n_td"]
        model = Aline(Embedder(2, 1, 32, 128, 3, "mix"), Encoder(32, 128, 4, 0.0, 3), OutputHead(2, 1, 32, 128)).cuda()
        batch = GPTask(dim_x=2, embedding_type="mix", n_context_init=1, n_query_init=40, n_target_theta=3, n_target_data=n_td,
                       device=torch.device("cuda")).sample_batch(B)
        mask_type = kw.get("mask", "all")
        if mask_type == "split":
            batch["target_mask"] = create_target_mask("split", "mix", n_td, 3, None, None, None, None, "data")
    model.set_precision("f16x3").train()
    with torch.no_grad():
        ro = Rollout(model, batch, T, select="sample", keep_acts=True)
        assert ro.path == "s3::step_kernel" and ro.saved_acts is not None
        ro.run()
        assert torch.isfinite(ro.saved_acts).all()
        terms = reinforce_terms(ro, emb, mask_type)
        grads = []
        for flags in ([], ["NO_BWD_SAVED_ACTS"]):
            with _lib.debug(*flags):
                for p in model.parameters():
                    p.grad = None
                backward(model, ro, terms["g_logp"], terms["g_ll"], t_chunk=kw.get("t_chunk"))
                torch.cuda.synchronize()
            grads.append({k: p.grad.clone() for k, p in model.named_parameters()})
    worst = ("", 0.0)
    floor = 1e-2 * max(float(g.abs().max()) for g in grads[1].values())
    for k in grads[0]:
        ref = grads[1][k]
        assert torch.isfinite(grads[0][k]).all(), k
        err = float((grads[0][k] - ref).abs().max()) / max(float(ref.abs().max()), floor)
        if err > worst[1]:
            worst = (k, err)
    assert worst[1] < 5e-4, worst


@pytest.mark.parametrize("T,B,n_td", [(12, 8, 100), (50, 4, 100), (20, 6, 40)])
def test_fused_backward_kernels_at_the_cfg3_key_counts(T, B, n_td):
    """BASELINE configs[2] (al_mix dx = 2: 100 target points + 3 theta tokens among the keys, split mask): up to 1 + 49 + 103 = 153
    keys per instance: beyond 48 keys the attention runs the per-op kernels (fp32 VALU attention + in-projection GEMMs) while the
    token-local tail, the acquisition head and the GMM heads stay on their fused kernels.  That mix against the all-per-op
    pipeline on the same rollout and upstream gradients: 115, 153 and 62 keys."""
    from aline_amd import Aline, Embedder, Encoder, OutputHead, _lib
    from aline_amd.rollout import Rollout
    from aline_amd.tasks import GPTask
    from aline_amd.train import backward, reinforce_terms
    from aline_amd.utils import create_target_mask
    torch.manual_seed(7)
    dev = torch.device("cuda")
    task = GPTask(dim_x=2, embedding_type="mix", n_context_init=1, n_query_init=200, n_target_theta=3, n_target_data=n_td, device=dev)
    batch = task.sample_batch(B)
    batch["target_mask"] = create_target_mask("split", "mix", n_td, 3, None, None, None, None, "data")
    model = Aline(Embedder(2, 1, 32, 128, 3, "mix"), Encoder(32, 128, 4, 0.0, 3), OutputHead(2, 1, 32, 128)).cuda()
    with torch.no_grad():
        ro = Rollout(model, batch, T, select="sample").run()
        terms = reinforce_terms(ro, "mix", "split")
        grads = []
        for flags in ([], ["NO_BWD_TAIL", "NO_BWD_ATTN_BLOCK", "NO_BWD_ACQ", "NO_BWD_LAYER_FWD", "NO_BWD_GMM_FUSED", "NO_BWD_GMM128",
                           "NO_BWD_GMM_BATCHED"]):
            with _lib.debug(*flags):
                for p in model.parameters():
                    p.grad = None
                backward(model, ro, terms["g_logp"], terms["g_ll"])
                torch.cuda.synchronize()
            grads.append({k: p.grad.clone() for k, p in model.named_parameters()})
    floor = 1e-2 * max(float(g.abs().max()) for g in grads[1].values())
    worst, errs = ("", 0.0), {}
    for k in grads[0]:
        ref = grads[1][k]
        assert torch.isfinite(grads[0][k]).all(), k
        err = errs[k] = float((grads[0][k] - ref).abs().max()) / max(float(ref.abs().max()), floor)
        if err > worst[1]:
            worst = (k, err)
    # exact-fp32 products in different summation orders.  With a few hundred instances ONE knife-edge ReLU (a hidden unit whose
    # pre-activation is within rounding of zero takes the other side in one of the two forwards: the gate-flip arbitration of
    # test_fused_backward_kernels_in_mix_mode_with_target_data_keys) moves an element of that unit's own gradients by up to 1 % and
    # everything upstream of it by ~1e-4: the first-layer parameters of the heads are held to 2e-2, everything else to 5e-4
    loose = max((e for k, e in errs.items() if k.startswith("head.") and (".0.weight" in k or ".0.bias" in k)), default=0.0)
    tight = max((e for k, e in errs.items() if not (k.startswith("head.") and (".0.weight" in k or ".0.bias" in k))), default=0.0)
    assert tight < 5e-4 and loose < 2e-2, (worst, tight, loose)


def test_train_step_graph_rollout_takes_the_new_batch():
    """`train_step` replays its sampled rollout from a HIP graph kept per (model, shapes, T): a second step on another batch
    must be the rollout of THAT batch -- same designs, log-probabilities and log-likelihoods as a fresh eager rollout with
    the uniform numbers the replay used -- and the flat gradient buffer must be what the optimiser sees."""
    from aline_amd import Aline, Embedder, Encoder, OutputHead
    from aline_amd.rollout import Rollout
    from aline_amd.tasks import HiddenLocation
    from aline_amd import train as tr
    torch.manual_seed(11)
    model = Aline(Embedder(2, 1, 32, 128, 2, "theta"), Encoder(32, 128, 4, 0.0, 3), OutputHead(2, 1, 32, 128)).cuda()
    task = HiddenLocation()
    T, B = 7, 12
    b1, b2 = task.sample_batch(B), task.sample_batch(B)
    tr.train_step(model, b1, T)
    terms, ro = tr.train_step(model, b2, T)                      # replay with b2 copied into the graph's inputs
    torch.cuda.synchronize()
    assert ro._graph is not None
    fresh = Rollout(model, b2, T, select="sample", uniform=ro.uniform.clone()).run()
    torch.cuda.synchronize()
    assert torch.equal(fresh.idx, ro.idx) and torch.equal(fresh.log_prob, ro.log_prob)
    assert torch.equal(fresh.target_ll, ro.target_ll)
    flat, _ = tr.flat_grads(model)
    off = 0
    for p in model.parameters():
        assert p.grad.data_ptr() == flat.data_ptr() + 4 * off
        off += p.numel()
    assert float(flat.abs().max()) <= 1.0 + 1e-6                 # inf-norm clip at 1.0 (train_aline.py:138)


def _oracle_gradients_with_gate_flips(sd, cpu, cfg, T, forced, mask, g_logp, g_ll, eps):
    """fp64 autograd of the oracle on sum(g_logp * log_prob) + sum(g_ll * target_ll): the gradient g0 (flat vector over all
    parameters, in sd order) and, for every ReLU gate whose pre-activation is within `eps` of zero, the change of that
    gradient when the gate is inverted."""
    import aline_oracle as orc

    def grad(flip):
        for v in sd.values():
            v.grad = None
        orc.reset_relu_calls()
        orc.RELU_PROBE, orc.RELU_FLIP = ([] if flip is None else None), flip
        try:
            ref = orc.rollout(sd, cpu, cfg, T, forced_idx=forced, mask_type=mask)
            probe = orc.RELU_PROBE
        finally:
            orc.RELU_PROBE = orc.RELU_FLIP = None
        obj = (torch.stack(ref["log_prob"], 1) * g_logp).sum() + (torch.stack(ref["target_ll"]) * g_ll).sum()
        obj.backward()
        return torch.cat([v.grad.reshape(-1) for v in sd.values()]).clone(), probe, ref

    g0, probe, ref = grad(None)
    edges = []
    for i, _tag, h in probe:
        for pos in torch.nonzero(h.abs() < eps):
            edges.append((i, tuple(int(p) for p in pos), h.shape))
    deltas = []
    for i, pos, shape in edges:
        m = torch.zeros(shape, dtype=torch.bool)
        m[pos] = True
        deltas.append(grad({i: m})[0] - g0)
    n_gates = sum(int(h.numel()) for _, _, h in probe)
    return g0, deltas, ref, n_gates


def _explain_with_gate_flips(res, deltas):
    """res - D c for the 0 / 1 assignment c of the knife-edge gates that explains the residual best: least squares rounded, then
    single-gate improvements (greedy, from the rounded solution and from "no gate flipped") -- nearly collinear deltas (two gates of one
    hidden unit) make the plain rounded least-squares solution unreliable.  Returns (remaining residual, c)."""
    if not deltas:
        return res, torch.zeros(0, dtype=torch.float64)
    D = torch.stack(deltas, 1)
    best = None
    c0 = (torch.linalg.lstsq(D, res.unsqueeze(1)).solution.squeeze(1) > 0.5).double()
    for c in (c0, torch.zeros_like(c0)):
        c = c.clone()
        r = res - D @ c
        improved = True
        while improved:
            improved = False
            for i in range(D.shape[1]):
                step = D[:, i] * (1.0 - 2.0 * c[i])            # effect of toggling gate i on D c
                r2 = r - step
                if float(r2.norm()) < float(r.norm()) * (1.0 - 1e-9):
                    r, c[i], improved = r2, 1.0 - c[i], True
        if best is None or float(r.norm()) < float(best[0].norm()):
            best = (r, c)
    return best


@pytest.mark.parametrize("mask,B", [("all", 10), ("split", 11)])
def test_fused_backward_kernels_in_mix_mode_with_target_data_keys(mask, B):
    """The fused attention kernels with target DATA rows among the keys and a target mask (model/encoder.py:83-126:
    the candidates see context + visible targets): al_mix task with 8 target points + 3 theta tokens, T = 9 (<= 32 keys).
    The fused kernels AND the per-op pipeline, each against the fp64 oracle's autograd (the arbiter) on the same designs and the
    same upstream gradients.  Knife-edge ReLUs: a gate whose pre-activation is within 5e-6 of zero in the oracle (fp32
    rounding of these O(1) pre-activations) may legitimately be taken either way by an fp32 forward, and one flipped gate moves
    every gradient upstream of it by that row's contribution (B = 10 is the batch where round 2 saw the two kernel sets 1 % apart
    on a GMM-bias element).  So the oracle is also differentiated with each such gate inverted, and a kernel's gradient must
    equal the oracle's for SOME assignment of those gates: g0 + sum_i c_i delta_i with every c_i in {0, 1} (least squares,
    rounded) -- nothing else is masked."""
    from aline_amd import Aline, Embedder, Encoder, OutputHead, _lib
    from aline_amd.rollout import Rollout
    from aline_amd.tasks import GPTask
    from aline_amd.train import backward, reinforce_terms
    from aline_amd.utils import create_target_mask
    torch.manual_seed(3)
    dev = torch.device("cuda")
    task = GPTask(dim_x=2, embedding_type="mix", n_context_init=2, n_query_init=40, n_target_theta=3, n_target_data=8, device=dev)
    batch = task.sample_batch(B)
    if mask == "split":
        batch["target_mask"] = create_target_mask("split", "mix", 8, 3, None, None, None, None, "data")
    model = Aline(Embedder(2, 1, 32, 128, 3, "mix"), Encoder(32, 128, 4, 0.0, 3), OutputHead(2, 1, 32, 128)).cuda()
    T = 9
    with torch.no_grad():
        ro = Rollout(model, batch, T, select="sample").run()
        terms = reinforce_terms(ro, "mix", mask)
        grads = []
        for flags in ([], ["NO_BWD_TAIL", "NO_BWD_ATTN_BLOCK", "NO_BWD_ACQ", "NO_BWD_LAYER_FWD", "NO_BWD_GMM_FUSED", "NO_BWD_GMM128",
                           "NO_BWD_GMM_BATCHED"]):
            with _lib.debug(*flags):
                for p in model.parameters():
                    p.grad = None
                backward(model, ro, terms["g_logp"], terms["g_ll"])
                torch.cuda.synchronize()
            grads.append({k: p.grad.cpu().double() for k, p in model.named_parameters()})
    sd = {k: v.detach().cpu().double().requires_grad_(True) for k, v in model.state_dict().items()}
    cpu = {k: (v.cpu().double() if v.is_floating_point() else v.cpu()) for k, v in batch.items() if torch.is_tensor(v)}
    cfg = dict(embedding_type="mix", n_head=4, num_layers=3, num_components=10, std_min=1e-4, n_target_theta=3)
    g0, deltas, ref, n_gates = _oracle_gradients_with_gate_flips(sd, cpu, cfg, T, ro.idx.cpu(), mask, terms["g_logp"].cpu().double(),
                                                              terms["g_ll"].cpu().double(), eps=5e-6)
    # the kernels' forward agrees with the oracle's (so the comparison below is about the backward)
    assert float((ro.target_ll.cpu().double() - torch.stack(ref["target_ll"]).detach()).abs().max()) < 1e-4
    assert len(deltas) <= 64 and len(deltas) < 1e-4 * n_gates, (len(deltas), n_gates)     # a handful of gates out of millions
    sizes = [v.numel() for v in sd.values()]
    floor = 1e-2 * float(g0.abs().max())
    for name, g in (("fused", grads[0]), ("per-op", grads[1])):
        flat = torch.cat([g[k].reshape(-1) for k in sd])
        res, flips = _explain_with_gate_flips(flat - g0, deltas)
        worst, off = ("", 0.0), 0
        for k, n in zip(sd, sizes):
            scale = max(float(g0[off:off + n].abs().max()), floor)
            err = float(res[off:off + n].abs().max()) / scale
            if err > worst[1]:
                worst = (k, err)
            off += n
        assert worst[1] < 2e-4, (name, worst, len(deltas), flips)


@pytest.mark.parametrize("prec", ["f32", "f16x3"])
@pytest.mark.parametrize("d,F,H", [(64, 256, 8), (128, 192, 4)])
def test_per_op_backward_of_wider_models_against_the_fp64_oracle(d, F, H, prec):
    """The backward of the model widths that have no fused kernels (what `d256.train_step` of the bench line runs: exact-fp32 GEMMs,
    LayerNorm / attention backward kernels at head_dim 8 and 32, `gmm_bwd_wide_kernel` at F > 128) against fp64 autograd of the oracle
    on the same designs and upstream gradients, with the knife-edge ReLU arbitration of the test above.  In f16x3 the forward
    recompute GEMMs of the backward run the forward's own 3-term f16 split (gradient products stay exact fp32)."""
    from aline_amd import Aline, Embedder, Encoder, OutputHead
    from aline_amd.rollout import Rollout
    from aline_amd.tasks import HiddenLocation
    from aline_amd.train import backward, reinforce_terms
    torch.manual_seed(21)
    model = Aline(Embedder(2, 1, d, F, 2, "theta"), Encoder(d, F, H, 0.0, 2), OutputHead(2, 1, d, F)).cuda().set_precision(prec)
    batch = HiddenLocation(n_query_init=30).sample_batch(6)
    T = 5
    with torch.no_grad():
        ro = Rollout(model, batch, T, select="sample").run()
        terms = reinforce_terms(ro, "theta", "all")
        backward(model, ro, terms["g_logp"], terms["g_ll"])
        torch.cuda.synchronize()
    g = {k: p.grad.cpu().double() for k, p in model.named_parameters()}
    sd = {k: v.detach().cpu().double().requires_grad_(True) for k, v in model.state_dict().items()}
    cpu = {k: (v.cpu().double() if v.is_floating_point() else v.cpu()) for k, v in batch.items() if torch.is_tensor(v)}
    cfg = dict(embedding_type="theta", n_head=H, num_layers=2, num_components=10, std_min=1e-4, n_target_theta=2)
    g0, deltas, ref, n_gates = _oracle_gradients_with_gate_flips(sd, cpu, cfg, T, ro.idx.cpu(), "all", terms["g_logp"].cpu().double(),
                                                              terms["g_ll"].cpu().double(), eps=5e-6)
    assert float((ro.target_ll.cpu().double() - torch.stack(ref["target_ll"]).detach()).abs().max()) < 1e-4
    assert len(deltas) <= 64, (len(deltas), n_gates)
    flat = torch.cat([g[k].reshape(-1) for k in sd])
    res, _flips = _explain_with_gate_flips(flat - g0, deltas)
    floor = 1e-2 * float(g0.abs().max())
    worst, off = ("", 0.0), 0
    for k, v in sd.items():
        n = v.numel()
        err = float(res[off:off + n].abs().max()) / max(float(g0[off:off + n].abs().max()), floor)
        if err > worst[1]:
            worst = (k, err)
        off += n
    assert worst[1] < 2e-4, (worst, len(deltas))


@pytest.mark.gpu
def test_weight_gradient_products_walked_loop_equals_the_general_loop():
    """`gemm_tn_block_kernel`: complete 64-row groups run a branch-free loop with walked row maps (identity maps and row groups of >= 64 rows),
    the rest the masked loop with per-row divisions.  Same gradients from both (the order of the fp32 sums differs: relative 1e-5), on a mix-mode
    model without fused backward kernels (d = 64, F = 256: per-op pipeline), 100 data targets + 2 theta tokens (the target rows are a 102-row group
    out of every 140-row instance: a non-identity map), B x T chosen so that the row counts are not multiples of 64."""
    from aline_amd import Aline, Embedder, Encoder, OutputHead, _lib
    from aline_amd.rollout import Rollout
    from aline_amd.tasks import GPTask
    from aline_amd.train import backward, reinforce_terms
    torch.manual_seed(5)
    dev = torch.device("cuda")
    model = Aline(Embedder(1, 1, 64, 256, 2, "mix"), Encoder(64, 256, 8, 0.0, 2), OutputHead(1, 1, 64, 256)).cuda().set_precision("f32")
    task = GPTask(dim_x=1, embedding_type="mix", n_context_init=1, n_query_init=37, n_target_theta=2, n_target_data=100, device=dev)
    batch = task.sample_batch(7)
    T = 3
    with torch.no_grad():
        ro = Rollout(model, batch, T, select="sample").run()
        terms = reinforce_terms(ro, "mix", "all")
    grads = []
    for flags in ((), ("NO_BWD_DW_WALK",)):
        for p in model.parameters():
            p.grad = None
        with _lib.debug(*flags), torch.no_grad():
            backward(model, ro, terms["g_logp"], terms["g_ll"])
            torch.cuda.synchronize()
        grads.append({k: p.grad.detach().clone() for k, p in model.named_parameters()})
    for k in grads[0]:
        a, b = grads[0][k], grads[1][k]
        scale = float(b.abs().max())
        assert scale > 0 or float(a.abs().max()) == 0, k
        assert float((a - b).abs().max()) <= 1e-5 * max(scale, 1e-30) + 1e-12, (k, float((a - b).abs().max()), scale)


@pytest.mark.parametrize("d,F,H,B,T", [(256, 1024, 8, 48, 5), (512, 256, 8, 20, 3)])
def test_scaled_f16_gradient_products_match_the_exact_fp32_products(d, F, H, B, T):
    """Round 4: in an F16X3 model at d >= 64 the gradient products of the per-op backward (dX = dY W on `gemm_nt_kernel<3>` with the
    gradient operand scaled by a per-tensor power of two; dW = dY^T X on `gemm_tn_f16_kernel`, 256 x 256 blocks, row chunks sized to
    whole rounds of workgroups, the scale words emitted by the producers of the gradient tensors) against the exact-fp32 products
    (`ALINE_DBG_BWD_GRAD_F32`) of the same rollout: every parameter gradient within 2e-4 of its max |grad| (the reference-autograd
    fixtures grad_cfg2_d256 / grad_cfg5_d512 hold the same kernels to the reference at 1e-3).  M = B T N is not a multiple of 32
    and spans several row chunks per block."""
    from aline_amd import Aline, Embedder, Encoder, OutputHead, _lib
    from aline_amd.rollout import Rollout
    from aline_amd.tasks import HiddenLocation
    from aline_amd.train import backward, reinforce_terms
    torch.manual_seed(3)
    model = Aline(Embedder(2, 1, d, F, 2, "theta"), Encoder(d, F, H, 0.0, 2), OutputHead(2, 1, d, F)).cuda().set_precision("f16x3").train()
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.02 * torch.randn_like(p))
    batch = HiddenLocation(n_query_init=60).sample_batch(B)
    grads = []
    with torch.no_grad():
        ro = Rollout(model, batch, T, select="sample").run()
        terms = reinforce_terms(ro, "theta", "all")
        for flags in ([], ["BWD_GRAD_F32"]):
            with _lib.debug(*flags):
                for p in model.parameters():
                    p.grad = None
                backward(model, ro, terms["g_logp"], terms["g_ll"], t_chunk=2 if flags == [] else None)
                torch.cuda.synchronize()
            grads.append({k: p.grad.clone() for k, p in model.named_parameters()})
    assert (B * T * 63) % 32 != 0
    worst = ("", 0.0)
    floor = 1e-2 * max(float(g.abs().max()) for g in grads[1].values())
    for k in grads[0]:
        ref = grads[1][k]
        assert torch.isfinite(grads[0][k]).all(), k
        err = float((grads[0][k] - ref).abs().max()) / max(float(ref.abs().max()), floor)
        if err > worst[1]:
            worst = (k, err)
    assert worst[1] < 2e-4, worst


@pytest.mark.parametrize("B,T,nq,task", [(37, 12, 150, "location"), (9, 40, 60, "location"), (6, 9, 40, "al_mix"),
                                         (4, 6, 40, "al_mix50"), (5, 10, 70, "al_mix100"), (3, 45, 60, "al_mix100")])
def test_f16_fused_backward_kernels_match_the_exact_fp32_fused_kernels(B, T, nq, task):
    """Round 4: the fused backward kernels of the d = 32 model on the f16 matrix pipe -- `tailbwd::tail16_kernel`, `acqb / gmmb::bwd16_kernel`,
    `abwd::attn_block_bwd16_kernel<2 | 3>`, and beyond 48 keys `abw8::attention_bwd8_kernel<4 | 7 | 10>` (head_dim 8, up to 160 keys): every group of four fp32 16x16x4 MFMAs a 3-term f16 split on `v_mfma_f32_16x16x16_f16`, the
    gradients scaled by the power of two of the upstream maximum (reduced by the producer kernels), weights packed x 2^8 -- against the
    exact-fp32 kernels they replace (`ALINE_DBG_BWD_GRAD_F32`) on the same rollout: two and three key tiles (T = 12 / 40), ragged token
    tiles, the al_mix geometry with data targets and a split mask.  Both sides recompute the hidden units of the heads and of the FFN from
    the same saved activations, one through the f16 split (1e-7 relative), one in fp32: a unit within that of zero changes side of its
    ReLU, and ONE such gate is 1e-4 .. 1e-3 of a head gradient at these row counts (millions of units: a handful per run).  So, as in the
    A/B test of the recompute switches below: nothing beyond 3e-3 of a tensor's max |grad|, four tensors in five within 2e-4."""
    from aline_amd import Aline, Embedder, Encoder, OutputHead, _lib
    from aline_amd.rollout import Rollout
    from aline_amd.train import backward, reinforce_terms
    torch.manual_seed(B + T)
    dev = torch.device("cuda")
    if task == "location":
        from aline_amd.tasks import HiddenLocation
        model = Aline(Embedder(2, 1, 32, 128, 2, "theta"), Encoder(32, 128, 4, 0.0, 3), OutputHead(2, 1, 32, 128))
        batch = HiddenLocation(n_query_init=nq, device=dev).sample_batch(B)
        emb, mask = "theta", "all"
    else:
        from aline_amd.tasks import GPTask
        from aline_amd.utils import create_target_mask
        model = Aline(Embedder(2, 1, 32, 128, 3, "mix"), Encoder(32, 128, 4, 0.0, 3), OutputHead(2, 1, 32, 128))
        ntd = 100 if task == "al_mix100" else 50 if task == "al_mix50" else 20      # 57 / 111 / 146 keys: `abw8::attention_bwd8_kernel<4 | 7 | 10>`
        batch = GPTask(dim_x=2, embedding_type="mix", n_context_init=1, n_query_init=nq, n_target_theta=3, n_target_data=ntd, device=dev).sample_batch(B)
        batch["target_mask"] = create_target_mask("split", "mix", ntd, 3, None, None, None, None, "data")
        emb, mask = "mix", "split"
    model = model.cuda().set_precision("f16x3").train()
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.05 * torch.randn_like(p))
    grads = []
    with torch.no_grad():
        ro = Rollout(model, batch, T, select="sample").run()
        assert ro.path == "s3::step_kernel", ro.path
        terms = reinforce_terms(ro, emb, mask)
        for flags in ([], ["BWD_GRAD_F32"]):
            with _lib.debug(*flags):
                for p in model.parameters():
                    p.grad = None
                backward(model, ro, terms["g_logp"], terms["g_ll"])
                torch.cuda.synchronize()
            grads.append({k: p.grad.clone() for k, p in model.named_parameters()})
    new, ref = grads
    floor = 1e-2 * max(float(g.abs().max()) for g in ref.values())
    errs = {}
    for k in ref:
        assert torch.isfinite(new[k]).all(), k
        errs[k] = float((new[k] - ref[k]).abs().max()) / max(float(ref[k].abs().max()), floor)
    worst = max(errs.items(), key=lambda kv: kv[1])
    assert worst[1] < 3e-3, worst
    close = sum(e < 2e-4 for e in errs.values())
    assert close >= 0.8 * len(errs), (close, len(errs), sorted(errs.items(), key=lambda kv: -kv[1])[:6])


@pytest.mark.parametrize("d,F,H,B,T,tc,mask", [(256, 512, 8, 24, 9, 4, False), (256, 256, 8, 5, 40, 16, False), (512, 128, 8, 9, 12, 5, True)])
def test_image_recompute_and_key_row_products_match_the_generic_recompute(d, F, H, B, T, tc, mask):
    """Round 4, second half: the per-op backward of a d = 256 / 512 F16X3 model recomputes the forward with the rollout's own layer kernel
    run over the (step, episode) instances (`x3::layer_save_kernel` / `x5::`: attention output, both LayerNorm inputs and outputs, hidden
    units and Q as fp32 rows; K / V of the key rows by a gather GEMM scattered to their token rows) and splits the in-projection's
    gradient products into a dense Q part and a K | V part over the key rows (`gemm_tn_f16_kernel` with an index list, `gemm_nt_kernel`
    with gather + scatter-add).  Against the generic recompute (`ALINE_DBG_NO_BWD_IMAGE_RECOMPUTE`) and the all-rows products
    (`ALINE_DBG_NO_BWD_KV_SPARSE`) of the same rollout, several chunks of different key counts (1 - 3 key tiles), a ragged last chunk,
    with and without a target mask."""
    from aline_amd import Aline, Embedder, Encoder, OutputHead, _lib
    from aline_amd.rollout import Rollout
    from aline_amd.tasks import HiddenLocation
    from aline_amd.train import backward, reinforce_terms
    from aline_amd.utils import create_target_mask
    torch.manual_seed(d + T)
    model = Aline(Embedder(2, 1, d, F, 2, "theta"), Encoder(d, F, H, 0.0, 2), OutputHead(2, 1, d, F)).cuda().set_precision("f16x3").train()
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.02 * torch.randn_like(p))
    batch = HiddenLocation(n_query_init=45).sample_batch(B)
    if mask:
        batch["target_mask"] = torch.tensor([True, False], device="cuda")
    grads = []
    with torch.no_grad():
        ro = Rollout(model, batch, T, select="sample").run()
        assert ro.path in ("x3::layer_kernel", "x5::layer_kernel"), ro.path
        terms = reinforce_terms(ro, "theta", "all")
        for flags in ([], ["NO_BWD_KV_SPARSE"], ["NO_BWD_IMAGE_RECOMPUTE", "NO_BWD_KV_SPARSE"]):
            with _lib.debug(*flags):
                for p in model.parameters():
                    p.grad = None
                backward(model, ro, terms["g_logp"], terms["g_ll"], t_chunk=tc)
                torch.cuda.synchronize()
            grads.append({k: p.grad.clone() for k, p in model.named_parameters()})
    ref = grads[2]
    floor = 1e-2 * max(float(g.abs().max()) for g in ref.values())
    for which in (0, 1):
        errs = {}
        for k in ref:
            assert torch.isfinite(grads[which][k]).all(), k
            errs[k] = float((grads[which][k] - ref[k]).abs().max()) / max(float(ref[k].abs().max()), floor)
        # The two recomputes round differently (1e-6): a hidden unit of a head within that of zero changes side, and with a few hundred
        # target rows ONE such ReLU gate is 1e-3 of a head gradient (tools/ab_backward_switches.py: the exact-fp32 recompute is as far
        # from either).  So: nothing beyond 3e-3 (the bound the reference-autograd fixtures use for fp32 against fp32), and four tensors
        # in five within 2e-4 -- a wrong activation or a wrong key row is an O(1) error in most of them.
        worst = max(errs.items(), key=lambda kv: kv[1])
        assert worst[1] < 3e-3, (which, worst)
        close = sum(e < 2e-4 for e in errs.values())
        assert close >= 0.8 * len(errs), (which, close, len(errs), sorted(errs.items(), key=lambda kv: -kv[1])[:6])


@pytest.mark.parametrize("d,F,H,B,T,nq", [(256, 256, 8, 6, 4, 50), (256, 256, 8, 5, 20, 40), (256, 256, 8, 3, 36, 60), (128, 192, 4, 7, 18, 30),
                                          (512, 128, 8, 4, 3, 40), (512, 128, 8, 3, 28, 50), (512, 128, 8, 2, 40, 50)])
def test_matrix_pipe_attention_backward_at_head_dim_32_and_64_matches_the_valu_kernel(d, F, H, B, T, nq):
    """`abww::attention_bwd_wide_kernel<32 | 64, NKT>` (attn_bwd_wide.h, round 4: the attention backward of the wide models on the fp32
    matrix pipe, one wave per head, dK / dV resident over the row tiles) against the fp32 VALU `attention_bwd_kernel` it replaces
    (`ALINE_DBG_NO_BWD_ATTN_MFMA`): every parameter gradient of the same rollout, exact-fp32 model (precision f32: no f16 scaling in the
    way), 1 / 2 / 3 key tiles at head_dim 32 (6, 22 and 38 keys; 4 heads too) and at head_dim 64 (5, 30 and 42 keys); token counts that are not
    multiples of 16 (ragged last row tile)."""
    from aline_amd import Aline, Embedder, Encoder, OutputHead, _lib
    from aline_amd.rollout import Rollout
    from aline_amd.tasks import HiddenLocation
    from aline_amd.train import backward, reinforce_terms
    torch.manual_seed(d + T)
    model = Aline(Embedder(2, 1, d, F, 2, "theta"), Encoder(d, F, H, 0.0, 2), OutputHead(2, 1, d, F)).cuda().set_precision("f32").train()
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.02 * torch.randn_like(p))
    batch = HiddenLocation(n_query_init=nq).sample_batch(B)
    grads = []
    with torch.no_grad():
        ro = Rollout(model, batch, T, select="sample").run()
        terms = reinforce_terms(ro, "theta", "all")
        for flags in ([], ["NO_BWD_ATTN_MFMA"]):
            with _lib.debug(*flags):
                for p in model.parameters():
                    p.grad = None
                backward(model, ro, terms["g_logp"], terms["g_ll"])
                torch.cuda.synchronize()
            grads.append({k: p.grad.clone() for k, p in model.named_parameters()})
    worst = ("", 0.0)
    floor = 1e-2 * max(float(g.abs().max()) for g in grads[1].values())
    for k in grads[0]:
        ref = grads[1][k]
        assert torch.isfinite(grads[0][k]).all(), k
        err = float((grads[0][k] - ref).abs().max()) / max(float(ref.abs().max()), floor)
        if err > worst[1]:
            worst = (k, err)
    assert worst[1] < 5e-5, worst       # the same fp32 products in another summation order

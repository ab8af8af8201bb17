"""CPU oracle for the ALINE amortized inference-and-design hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package (`aline_amd/`) may import this
module; only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` do,
and there only as the checker / the timed CPU baseline.

This is a from-scratch functional restatement (torch CPU tensors, fp32 by default, fp64 on
request) of the reference's algorithm for the path
    Embedder.forward -> Encoder.forward -> OutputHead.forward (+ compute_ll, update_batch,
    the REINFORCE reductions and the sequential EIG bounds).
It takes a plain `state_dict` (same key names as the reference) and plain
dict batches.  Each function cites the reference file:line it restates (paths relative to
/root/reference).

PARITY IS PINNED: `oracle/make_golden.py` imports the reference itself in the build container
and writes `tests/golden/*.npz`; `tests/test_oracle_golden.py` checks every function below
against those fixtures.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch

Tensor = torch.Tensor
LN_EPS = 1e-5  # torch TransformerEncoderLayer default layer_norm_eps


# ----------------------------------------------------------------------------------------
# small helpers
# ----------------------------------------------------------------------------------------
def _lin(x: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
    y = x @ w.t()
    return y if b is None else y + b


# Test probes of the ReLUs (tests/test_backward_gpu.py).  Two correct fp32 forwards may take different sides of a ReLU whose
# pre-activation is within rounding of zero, and ONE flipped gate changes every gradient upstream of it (the row's whole
# contribution through that unit).  RELU_PROBE (a list): every ReLU appends (call index, key prefix of its Linear,
# pre-activation).  RELU_FLIP ({call index: bool mask}): the gates under the mask are inverted (forward h * gate, so the value
# stays continuous) -- the test differentiates the oracle under the flips of the knife-edge gates to span what a correct
# kernel may return.  reset_relu_calls() restarts the call numbering before a rollout.
RELU_PROBE: Optional[list] = None
RELU_FLIP: Optional[dict] = None
_relu_calls = 0


def reset_relu_calls() -> None:
    global _relu_calls
    _relu_calls = 0


def _relu(h: Tensor, tag: str) -> Tensor:
    global _relu_calls
    i = _relu_calls
    _relu_calls += 1
    if RELU_PROBE is not None:
        RELU_PROBE.append((i, tag, h.detach()))
    if RELU_FLIP is not None and i in RELU_FLIP:
        gate = (h.detach() > 0) ^ RELU_FLIP[i]
        return h * gate.to(h.dtype)
    return torch.relu(h)


def _mlp(sd: Dict[str, Tensor], prefix: str, x: Tensor) -> Tensor:
    """nn.Sequential(Linear, ReLU, Linear) with keys `{prefix}.0.*`, `{prefix}.2.*`."""
    h = _relu(_lin(x, sd[f"{prefix}.0.weight"], sd[f"{prefix}.0.bias"]), f"{prefix}.0")
    return _lin(h, sd[f"{prefix}.2.weight"], sd[f"{prefix}.2.bias"])


def _layer_norm(x: Tensor, w: Tensor, b: Tensor) -> Tensor:
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)  # biased, as torch LayerNorm
    return (x - mu) / torch.sqrt(var + LN_EPS) * w + b


def cast_state_dict(sd: Dict[str, Tensor], dtype=torch.float32) -> Dict[str, Tensor]:
    return {k: v.detach().to("cpu", dtype) for k, v in sd.items()}


# ----------------------------------------------------------------------------------------
# a2  Embedder.forward                      model/embedder.py:67-95, :97-126, :128-168, :170-214
# ----------------------------------------------------------------------------------------
def embed(sd, batch, embedding_type: str) -> Tensor:
    """Returns [B, N, d] in the order ctx | query | target_x | theta tokens."""
    cx, cy, qx = batch["context_x"], batch["context_y"], batch["query_x"]
    B, n_c = cx.shape[0], cx.shape[1]
    if embedding_type == "theta":
        x_all = torch.cat([cx, qx], dim=1)                      # embedder.py:144-149
    elif embedding_type in ("data", "mix"):
        x_all = torch.cat([cx, qx, batch["target_x"]], dim=1)   # embedder.py:113-118, :188-193
    else:
        raise ValueError(f"Unknown embedding type: {embedding_type}")
    xe = _mlp(sd, "embedder.x_embedder", x_all)
    ye = _mlp(sd, "embedder.y_embedder", cy)                    # context y only (embedder.py:153)
    parts = [xe[:, :n_c] + ye, xe[:, n_c:]]
    if embedding_type in ("theta", "mix"):                      # embedder.py:162, :205
        tok = sd["embedder.theta_tokens"]
        parts.append(tok.unsqueeze(0).expand(B, -1, -1))
    return torch.cat(parts, dim=1)


# ----------------------------------------------------------------------------------------
# a3  Encoder.create_mask                                        model/encoder.py:83-126
# ----------------------------------------------------------------------------------------
def allowed_keys(n_c: int, n_q: int, n_t: int, target_mask: Optional[Tensor]) -> Tensor:
    """Boolean [N, N]: allowed[i, j] == (reference additive mask[i, j] == 0)."""
    N = n_c + n_q + n_t
    allowed = torch.zeros(N, N, dtype=torch.bool)
    allowed[:, :n_c] = True                                     # encoder.py:107
    if target_mask is not None:
        sel = torch.where(target_mask.to(torch.bool))[0] + n_c + n_q   # encoder.py:115-118
        allowed[n_c:n_c + n_q, sel] = True                      # encoder.py:121
    else:
        allowed[n_c:n_c + n_q, n_c + n_q:] = True               # encoder.py:124
    return allowed


# ----------------------------------------------------------------------------------------
# a4/a5  Encoder.forward = L x post-norm TransformerEncoderLayer      model/encoder.py:128-141
# (train path encoder.py:8-46 and the eval fast path compute the same masked attention)
# ----------------------------------------------------------------------------------------
def encoder(sd, x: Tensor, allowed: Tensor, n_head: int, num_layers: int,
            return_all: bool = False):
    B, N, d = x.shape
    hd = d // n_head
    neg = torch.zeros(N, N, dtype=x.dtype).masked_fill(~allowed, float("-inf"))
    outs = []
    for l in range(num_layers):
        p = f"encoder.encoder.layers.{l}"
        qkv = _lin(x, sd[f"{p}.self_attn.in_proj_weight"], sd[f"{p}.self_attn.in_proj_bias"])
        q, k, v = qkv.split(d, dim=-1)
        q = q.view(B, N, n_head, hd).transpose(1, 2)
        k = k.view(B, N, n_head, hd).transpose(1, 2)
        v = v.view(B, N, n_head, hd).transpose(1, 2)
        s = (q @ k.transpose(-1, -2)) / math.sqrt(hd) + neg
        a = torch.softmax(s, dim=-1) @ v                        # [B, H, N, hd]
        a = a.transpose(1, 2).reshape(B, N, d)
        sa = _lin(a, sd[f"{p}.self_attn.out_proj.weight"], sd[f"{p}.self_attn.out_proj.bias"])
        x = _layer_norm(x + sa, sd[f"{p}.norm1.weight"], sd[f"{p}.norm1.bias"])
        ff = _lin(_relu(_lin(x, sd[f"{p}.linear1.weight"], sd[f"{p}.linear1.bias"]), f"{p}.linear1"),
                  sd[f"{p}.linear2.weight"], sd[f"{p}.linear2.bias"])
        x = _layer_norm(x + ff, sd[f"{p}.norm2.weight"], sd[f"{p}.norm2.bias"])
        outs.append(x)
    return (x, outs) if return_all else x


# ----------------------------------------------------------------------------------------
# a8  GMMTargetHead.forward / _map_raw_output                 model/head.py:152-186, :251-266
# ----------------------------------------------------------------------------------------
def gmm_head(sd, z: Tensor, num_components: int, std_min: float = 1e-4):
    """z [B, tok, d] -> (means, stds, weights) each [B, tok, C] (dim_y == 1)."""
    outs = [_mlp(sd, f"head.target_head.heads.{c}", z) for c in range(num_components)]
    # stack -> [C, B, tok, 3] -> movedim -> [B, tok, 3, C] -> flatten -> [B, tok, 3*C]
    cat = torch.stack(outs).movedim(0, -1).flatten(-2, -1)      # head.py:264
    raw_mean, raw_std, raw_w = torch.chunk(cat, 3, dim=-1)      # head.py:265
    return (raw_mean,
            torch.nn.functional.softplus(raw_std) + std_min,    # head.py:176
            torch.softmax(raw_w, dim=-1))                       # head.py:177


# ----------------------------------------------------------------------------------------
# a9  compute_ll                                 utils/eval.py:200-207 (== model/head.py:232-249)
# ----------------------------------------------------------------------------------------
def compute_ll(value: Tensor, means: Tensor, stds: Tensor, weights: Tensor) -> Tensor:
    """value [B, n_t, 1] broadcast against [B, n_t, C] -> [B, n_t]."""
    logp = (-((value - means) ** 2) / (2 * stds ** 2) - torch.log(stds)
            - math.log(math.sqrt(2 * math.pi)))
    return torch.logsumexp(logp + torch.log(weights), dim=-1)


# ----------------------------------------------------------------------------------------
# a6/a7  AcquisitionHead + OutputHead.forward                model/head.py:35-44, :319-393
# ----------------------------------------------------------------------------------------
def head(sd, z: Tensor, n_c: int, n_q: int, num_components: int = 10, std_min: float = 1e-4,
         forced_idx: Optional[Tensor] = None, time_t: Optional[Tensor] = None,
         with_query_gmm: bool = True):
    """forced_idx [B] (teacher forcing) replaces Categorical.sample(); None = eval argmax."""
    zq, zt_ = z[:, n_c:n_c + n_q], z[:, n_c + n_q:]
    if time_t is not None:                                      # head.py:342-345
        B = z.shape[0]
        zq_in = torch.cat([zq, time_t.reshape(1, 1, 1).expand(B, n_q, 1).to(z.dtype)], dim=-1)
    else:
        zq_in = zq
    logits = _mlp(sd, "head.acquisition_head.predictor", zq_in).squeeze(-1)   # head.py:27-33
    zt = torch.softmax(logits, dim=-1)
    if forced_idx is None:                                      # head.py:355-358
        prob, idx = torch.max(zt, -1)
        log_prob = torch.log(prob)
    else:                                                       # head.py:350-354
        idx = forced_idx.reshape(-1).long()
        # Categorical(probs).log_prob: log of the normalised, clamped probabilities
        probs = zt / zt.sum(-1, keepdim=True)
        eps = torch.finfo(probs.dtype).eps
        logits_n = torch.log(probs.clamp(min=eps, max=1 - eps))
        log_prob = logits_n.gather(1, idx[:, None]).squeeze(1)
    out = {
        "idx": idx[:, None], "log_prob": log_prob, "zt": zt,
        "posterior": gmm_head(sd, zt_, num_components, std_min),
    }
    if with_query_gmm:                                          # head.py:366
        out["posterior_query"] = gmm_head(sd, zq, num_components, std_min)
    return out


# ----------------------------------------------------------------------------------------
# a1  Aline.forward                                                   model/base.py:32-50
# ----------------------------------------------------------------------------------------
def forward(sd, batch, cfg, forced_idx=None, with_query_gmm=True, return_hidden=False):
    """cfg: dict(embedding_type, n_head, num_layers, num_components, std_min, time_token)."""
    n_c, n_q = batch["context_x"].shape[1], batch["query_x"].shape[1]
    n_t = batch["target_all"].shape[1]
    emb = embed(sd, batch, cfg["embedding_type"])
    allowed = allowed_keys(n_c, n_q, n_t, batch.get("target_mask"))
    z = encoder(sd, emb, allowed, cfg["n_head"], cfg["num_layers"])
    t = batch.get("t") if cfg.get("time_token") else None
    out = head(sd, z, n_c, n_q, cfg.get("num_components", 10), cfg.get("std_min", 1e-4),
               forced_idx=forced_idx, time_t=t, with_query_gmm=with_query_gmm)
    if return_hidden:
        out["embedding"], out["encoding"] = emb, z
    return out


# ----------------------------------------------------------------------------------------
# a12  Task.update_batch                                          tasks/base_task.py:103-154
# ----------------------------------------------------------------------------------------
def update_batch(batch, idx: Tensor):
    """Moves query `idx[b]` (index into the current compacted list) to the end of the context."""
    B, n_q = batch["query_x"].shape[:2]
    ar = torch.arange(B)
    i = idx.reshape(-1)
    keep = torch.ones(B, n_q, dtype=torch.bool)
    keep[ar, i] = False
    new = dict(batch)
    for k in ("x", "y"):
        q = batch[f"query_{k}"]
        new[f"context_{k}"] = torch.cat([batch[f"context_{k}"], q[ar, i][:, None]], dim=1)
        new[f"query_{k}"] = q[keep].view(B, n_q - 1, q.shape[-1])
    return new


# ----------------------------------------------------------------------------------------
# a13  select_targets_by_mask                                 utils/target_mask.py:107-125
# ----------------------------------------------------------------------------------------
def select_targets_by_mask(res: Tensor, target_mask: Tensor) -> Tensor:
    return res[:, torch.where(target_mask.to(torch.bool))[0]]


def create_target_mask(mask_type, embedding_type, n_target_data, n_target_theta,
                       predefined_mask=None, attend_to=None, selected=None) -> Tensor:
    """Deterministic part of utils/target_mask.py:5-104 (random choices passed in explicitly)."""
    n_t = n_target_data + n_target_theta
    m = torch.zeros(n_t, dtype=torch.bool)
    if mask_type == "all":
        m[:] = True
    elif mask_type == "none":
        pass
    elif mask_type == "partial":
        m[torch.as_tensor(selected, dtype=torch.long)] = True
    elif mask_type == "predefined":
        for i, on in enumerate(predefined_mask):
            if i < n_t and on:
                m[i] = True
    elif mask_type == "split":
        if embedding_type == "mix":
            if attend_to == "data":
                m[:n_target_data] = True
            else:
                m[n_target_data:] = True
    return m


# ----------------------------------------------------------------------------------------
# a10/a11  loop body + REINFORCE reductions                         train_aline.py:80-125
# ----------------------------------------------------------------------------------------
def step_nlls(target_ll: Tensor, target_mask: Optional[Tensor], embedding_type: str,
              mask_type: str, n_target_theta: int):
    """Returns (nll_for_query [B], nll [B]) as train_aline.py:97-110."""
    n_t = target_ll.shape[1]
    tm = target_mask if target_mask is not None else torch.ones(n_t, dtype=torch.bool)
    masked = select_targets_by_mask(target_ll, tm)
    if embedding_type == "mix" and mask_type == "all":
        nll_q = -(masked[:, :-n_target_theta].mean(-1) + masked[:, -n_target_theta:].mean(-1))
    else:
        nll_q = -masked.mean(-1)
    if embedding_type == "mix":
        nll = -(target_ll[:, :-n_target_theta].mean(-1) + target_ll[:, -n_target_theta:].mean(-1))
    else:
        nll = -target_ll.mean(-1)
    return nll_q, nll


def reinforce_losses(log_probs: Tensor, nlls_q: List[Tensor], nlls: List[Tensor],
                     gamma: float = 1.0):
    """log_probs [B, T]; returns (R [B, T-1], design_loss, predict_loss)  train_aline.py:113-125."""
    T = log_probs.shape[1]
    R = torch.stack([(gamma ** t) * torch.clamp(nlls_q[t - 1] - nlls_q[t], min=0.0)
                     for t in range(1, T)], 1)
    R = (R - R.mean(0, keepdim=True)) / (R.std(0, keepdim=True) + 1e-9)   # unbiased std
    design_loss = -torch.mean(log_probs[:, :-1] * R)
    predict_loss = torch.mean(torch.stack(nlls))
    return R, design_loss, predict_loss


def rollout(sd, batch, cfg, T: int, forced_idx: Optional[Tensor] = None,
            mask_type: str = "all", with_query_gmm: bool = False, time_schedule: str = "train"):
    """T-step acquisition loop (train_aline.py:80-110 / utils/eval.py:24-30).

    forced_idx [B, T] teacher-forces the designs; None = eval-mode argmax.
    time_schedule (time-token models): "train" feeds t / T (train_aline.py:82), "eval" feeds (T - t) / T as the
    reference's evaluation loop does (utils/eval.py:24).
    Returns per-step lists and the final batch.
    """
    res = {"idx": [], "log_prob": [], "zt": [], "target_ll": [], "nll_q": [], "nll": [],
           "means": [], "stds": [], "weights": []}
    for t in range(T):
        if cfg.get("time_token"):
            batch = dict(batch)
            batch["t"] = torch.tensor([(T - t) / T if time_schedule == "eval" else t / T])
        out = forward(sd, batch, cfg, None if forced_idx is None else forced_idx[:, t],
                      with_query_gmm=with_query_gmm)
        batch = update_batch(batch, out["idx"])
        m, s, w = out["posterior"]
        ll = compute_ll(batch["target_all"], m, s, w)
        nq, n = step_nlls(ll, batch.get("target_mask"), cfg["embedding_type"], mask_type,
                          cfg.get("n_target_theta", 0))
        for k, v in (("idx", out["idx"]), ("log_prob", out["log_prob"]), ("zt", out["zt"]),
                     ("target_ll", ll), ("nll_q", nq), ("nll", n), ("means", m), ("stds", s),
                     ("weights", w)):
            res[k].append(v)
    res["batch"] = batch
    return res


# ----------------------------------------------------------------------------------------
# a15  HiddenLocation.log_likelihood / total_density        tasks/location_finding.py:110-164
# ----------------------------------------------------------------------------------------
def location_log_likelihood(y: Tensor, xi: Tensor, theta: Tensor, noise_scale: float = 0.5,
                            base_signal: float = 0.1, max_signal: float = 1e-4) -> Tensor:
    """y [.., 1], xi [.., D], theta [.., K, D] -> [.., 1]."""
    sq = (xi.unsqueeze(-2).expand(theta.shape) - theta).pow(2).sum(-1)   # location_finding.py:121-123
    signal = torch.log(base_signal + (max_signal + sq).pow(-1).sum(-1, keepdim=True))
    var = noise_scale ** 2
    return (-((y - signal) ** 2) / (2 * var) - math.log(noise_scale)
            - math.log(math.sqrt(2 * math.pi)))


def location_forward_signal(xi: Tensor, theta: Tensor, base_signal=0.1, max_signal=1e-4):
    sq = (xi.unsqueeze(-2).expand(theta.shape) - theta).pow(2).sum(-1)
    return torch.log(base_signal + (max_signal + sq).pow(-1).sum(-1, keepdim=True))


# ----------------------------------------------------------------------------------------
# a16  CESTask.log_likelihood + CensoredSigmoidNormal.log_prob
#      tasks/ces.py:96-115, :169-210; distributions/censored_sigmoid_normal.py:47-86
# ----------------------------------------------------------------------------------------
def _normal_cdf(x, loc, scale):
    return 0.5 * (1 + torch.erf((x - loc) / (scale * math.sqrt(2.0))))


def _normal_log_prob(x, loc, scale):
    return -((x - loc) ** 2) / (2 * scale ** 2) - torch.log(scale) - math.log(math.sqrt(2 * math.pi))


def _logit_clamped(y):
    """torch SigmoidTransform._inverse: clamp to [tiny, 1-eps] then log(y) - log1p(-y)."""
    fi = torch.finfo(y.dtype)
    y = y.clamp(min=fi.tiny, max=1.0 - fi.eps)
    return y.log() - (-y).log1p()


def _sigmoid_normal_log_prob(value, loc, scale):
    """TransformedDistribution(Normal(loc, scale), SigmoidTransform()).log_prob(value)."""
    x = _logit_clamped(value)
    ladj = -torch.nn.functional.softplus(-x) - torch.nn.functional.softplus(x)
    return _normal_log_prob(x, loc, scale) - ladj


def censored_sigmoid_normal_log_prob(value, loc, scale, lower, upper):
    """distributions/censored_sigmoid_normal.py:47-86 (without the host-syncing NaN raise)."""
    value, loc, scale = torch.broadcast_tensors(value, loc, scale)
    lower = torch.as_tensor(lower, dtype=value.dtype).expand_as(value)
    upper = torch.as_tensor(upper, dtype=value.dtype).expand_as(value)
    log_prob = _sigmoid_normal_log_prob(value, loc, scale)               # csn.py:54
    upper_cdf = 1.0 - _normal_cdf(_logit_clamped(upper), loc, scale)     # csn.py:57
    lower_cdf = _normal_cdf(_logit_clamped(lower), loc, scale)           # csn.py:58
    crit = 2 * torch.finfo(value.dtype).tiny                             # csn.py:60
    z_up = (_logit_clamped(upper) - loc) / scale                         # csn.py:65
    z_lo = (_logit_clamped(lower) - loc) / scale
    asym_up = _sigmoid_normal_log_prob(upper, loc, scale) - (crit + z_up.abs()).log()   # csn.py:68
    asym_lo = _sigmoid_normal_log_prob(lower, loc, scale) - (crit + z_lo.abs()).log()
    up_lc = torch.where(upper_cdf < crit, asym_up, upper_cdf.log())      # csn.py:71-75
    lo_lc = torch.where(lower_cdf < crit, asym_lo, lower_cdf.log())
    ninf = torch.full_like(log_prob, float("-inf"))
    log_prob = torch.where(value == upper, up_lc, log_prob)              # csn.py:78-81
    log_prob = torch.where(value == lower, lo_lc, log_prob)
    log_prob = torch.where(value > upper, ninf, log_prob)
    log_prob = torch.where(value < lower, ninf, log_prob)
    return log_prob


def ces_log_likelihood(y, xi, theta, noise_scale: float = 0.005, epsilon: float = 2.0 ** -22):
    """tasks/ces.py:169-210 (+ utility :96-115).  y [1,B,1], xi [1,B,6], theta [L,B,5] -> [L,B,1]."""
    rho, alpha, u = theta[..., 0:1], theta[..., 1:4], torch.exp(theta[..., 4:5])
    xi = torch.clamp(xi, min=0.01, max=100.0)
    b1, b2 = xi[..., :3], xi[..., 3:]

    def util(x):
        return torch.sum(alpha * x ** rho, dim=-1, keepdim=True) ** (1.0 / rho)

    mu = (util(b1) - util(b2)) * u
    sigma = (1 + torch.norm(b1 - b2, dim=-1, p=2, keepdim=True)) * noise_scale * u
    return censored_sigmoid_normal_log_prob(y, mu, sigma, epsilon, 1 - epsilon)


# ----------------------------------------------------------------------------------------
# a14  EIGStepLoss.step/forward + compute_EIG_from_history
#      loss/eig.py:174-209; utils/eval.py:42-80
# ----------------------------------------------------------------------------------------
def eig_bounds_from_history(log_lik_fn, theta_0: Tensor, x: Tensor, y: Tensor, thetas: Tensor,
                            stepwise: bool = False):
    """thetas [L, B, ...] contrastive samples (theta_0 is prepended as row 0).

    Returns (pce_bound, nmc_bound, seq_logprobs[L+1, B]); bounds [B] or [B, T] if stepwise.
    """
    L = thetas.shape[0]
    th = torch.cat([theta_0.unsqueeze(0), thetas], dim=0)              # eval.py:62
    S = torch.zeros(L + 1, x.shape[0], dtype=x.dtype)
    pces, nmcs = [], []
    for t in range(x.shape[1]):
        lp = log_lik_fn(y[:, t].unsqueeze(0), x[:, t].unsqueeze(0), th).squeeze(-1)   # eig.py:186-189
        S = S + lp                                                      # eig.py:192
        pces.append(S.logsumexp(0) - S[0])                              # eig.py:200
        nmcs.append(S[1:].logsumexp(0) - S[0])                          # eig.py:202
    if stepwise:
        pce, nmc = torch.stack(pces, -1), torch.stack(nmcs, -1)
    else:
        pce, nmc = pces[-1], nmcs[-1]
    return math.log(L + 1) - pce, math.log(L) - nmc, S                  # eval.py:77-78


# ----------------------------------------------------------------------------------------
# deterministic weights shared by the fixture generator and the tests (NOT reference code)
# ----------------------------------------------------------------------------------------
def state_dict_shapes(dim_x, dim_y, d, F, n_head, L, C, n_theta, embedding_type,
                      time_token=False):
    s = {}
    if embedding_type in ("theta", "mix"):
        s["embedder.theta_tokens"] = (n_theta, d)
    for nm, din in (("x", dim_x), ("y", dim_y)):
        s[f"embedder.{nm}_embedder.0.weight"] = (F, din)
        s[f"embedder.{nm}_embedder.0.bias"] = (F,)
        s[f"embedder.{nm}_embedder.2.weight"] = (d, F)
        s[f"embedder.{nm}_embedder.2.bias"] = (d,)
    for l in range(L):
        p = f"encoder.encoder.layers.{l}"
        s[f"{p}.self_attn.in_proj_weight"] = (3 * d, d)
        s[f"{p}.self_attn.in_proj_bias"] = (3 * d,)
        s[f"{p}.self_attn.out_proj.weight"] = (d, d)
        s[f"{p}.self_attn.out_proj.bias"] = (d,)
        s[f"{p}.linear1.weight"] = (F, d)
        s[f"{p}.linear1.bias"] = (F,)
        s[f"{p}.linear2.weight"] = (d, F)
        s[f"{p}.linear2.bias"] = (d,)
        for n in ("norm1", "norm2"):
            s[f"{p}.{n}.weight"] = (d,)
            s[f"{p}.{n}.bias"] = (d,)
    da = d + 1 if time_token else d
    s["head.acquisition_head.predictor.0.weight"] = (F, da)
    s["head.acquisition_head.predictor.0.bias"] = (F,)
    s["head.acquisition_head.predictor.2.weight"] = (1, F)
    s["head.acquisition_head.predictor.2.bias"] = (1,)
    for c in range(C):
        s[f"head.target_head.heads.{c}.0.weight"] = (F, d)
        s[f"head.target_head.heads.{c}.0.bias"] = (F,)
        s[f"head.target_head.heads.{c}.2.weight"] = (3 * dim_y, F)
        s[f"head.target_head.heads.{c}.2.bias"] = (3 * dim_y,)
    return s


def make_state_dict(seed: int, **dims) -> Dict[str, Tensor]:
    """Deterministic non-trivial weights: N(0, 1/fan_in)-ish matrices, non-zero biases,
    LayerNorm gains around 1.  Same generator order on every machine (torch CPU Philox-free
    mt19937 generator is platform-stable for a fixed torch build)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shp in state_dict_shapes(**dims).items():
        if k.endswith("norm1.weight") or k.endswith("norm2.weight"):
            sd[k] = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif k.endswith("bias"):
            sd[k] = 0.1 * torch.randn(shp, generator=g)
        elif k.endswith("theta_tokens"):
            sd[k] = torch.randn(shp, generator=g)
        else:
            fan_in = shp[-1]
            sd[k] = torch.randn(shp, generator=g) * (1.5 / math.sqrt(fan_in))
    return sd

"""Mirror of the reference's utils/eval.py for the hot path: compute_ll (:200-207), get_traces
(:8-39) and compute_EIG_from_history (:42-80), all on the HIP kernels."""
import math

import torch

from .. import _lib
from ..loss.eig import EIGStepLoss


def compute_ll(value, means, stds, weights):
    """GMM log-likelihood logsumexp_c(Normal(mu_c, sd_c).log_prob(v) + log w_c)  (eval.py:200-207).
    value [B, n_t, 1] (or [B, n_t]) against [B, n_t, C] -> [B, n_t]."""
    if torch.is_grad_enabled() and any(t.requires_grad for t in (means, stds, weights) if torch.is_tensor(t)):
        raise NotImplementedError(
            "aline_amd.utils.compute_ll is the forward-only HIP kernel; for autograd use the reference's own "
            "torch expression (utils/eval.py:200-207) on the outputs of Aline.forward, or aline_amd.train")
    C_ = means.shape[-1]
    lead = means.shape[:-1]
    v = _lib.f32(value).expand(*lead, 1) if value.dim() == means.dim() else _lib.f32(value)
    v = v.reshape(-1).contiguous()
    m, s, w = (_lib.f32(t).reshape(-1, C_) for t in (means, stds, weights))
    out = torch.empty(v.shape[0], dtype=torch.float32, device=m.device)
    _lib.check(_lib.lib.aline_compute_ll(v.data_ptr(), m.data_ptr(), s.data_ptr(), w.data_ptr(),
                                         v.shape[0], C_, out.data_ptr(), _lib.stream_ptr(m.device)),
               "compute_ll")
    return out.reshape(lead)


@torch.no_grad()
def get_traces(model, experiment, T=30, batch_size=40, time_token=False):
    """Eval-mode rollout (eval.py:8-39) on the static-slot rollout API: returns theta_0, designs x
    [B, n_ctx0+T, dx] (unnormalised) and outcomes y in order of acquisition."""
    from ..rollout import Rollout
    model.eval()
    theta_shape = experiment.sample_theta((batch_size)).shape
    batch = experiment.sample_batch(batch_size)
    # the reference's eval loop feeds batch.t = (T - t) / T (eval.py:24), not the t / T of training
    ro = Rollout(model, batch, T, select="argmax", time_token_T=T if time_token else 0, time_token_reverse=bool(time_token))
    ro.run_checked()        # (f16x3: falls back to f32 if an operand left f16's range)
    cx, cy = ro.export_context()
    theta_0 = batch.target_theta.reshape(*theta_shape)
    return theta_0, experiment.unnormalise_design(cx), cy


@torch.no_grad()
def compute_EIG_from_history(experiment, theta_0, x, y, L=int(1e6), batch_size=40, stepwise=False, thetas=None, fused=True):
    """sPCE / sNMC bounds from a design history (eval.py:42-80).  `thetas` [L, B, ...] replaces the contrastive draw of
    eval.py:61 (tests pin the function to the reference's bounds on the reference's own draw).  fused=False: the reference's own
    structure, one `EIGStepLoss` step per design (what tasks without a whole-history kernel use)."""
    T = x.shape[1]
    if thetas is None:
        thetas = experiment.sample_theta((L, batch_size))
    thetas = torch.concat([theta_0.unsqueeze(0), thetas], dim=0).contiguous()
    if fused and hasattr(experiment, "native_eig_history"):
        # all T steps in one pass over thetas (the T step launches read thetas and the running sums T times: eig.h)
        out = experiment.native_eig_history(thetas, x, y)
        if out is not None:      # (None: the task's history kernel does not take this shape -- the step kernels below)
            pce, nmc = out
            return (pce, nmc) if stepwise else (pce[:, -1], nmc[:, -1])
    criterion = EIGStepLoss(L, batch_size, experiment, reduction="none", device=x.device)
    pce_l, nmc_l = [], []
    for t in range(T):
        last = t == T - 1
        if stepwise or last:
            pce, nmc = criterion(y[:, t], x[:, t], thetas)
            pce_l.append(pce)
            nmc_l.append(nmc)
        else:
            criterion.step(y[:, t], x[:, t], thetas)   # eval.py:73-74 recomputes the LSE; only the last is used
    if stepwise:
        pce, nmc = torch.stack(pce_l, dim=-1), torch.stack(nmc_l, dim=-1)
    else:
        pce, nmc = pce_l[-1], nmc_l[-1]
    return math.log(L + 1) - pce, math.log(L) - nmc


def calculate_gmm_variance(mixture_means, mixture_stds, mixture_weights):
    """Variance of the per-point GMM prediction (utils/misc.py:244-279), the uncertainty-sampling score computed from
    `posterior_out_query`: sum_c w_c (sd_c^2 + (mu_c - sum_c w_c mu_c)^2).  [B, n, C] (weights may be [B, C]) -> [B, n]."""
    w = mixture_weights.unsqueeze(1).expand_as(mixture_means) if mixture_weights.dim() == 2 else mixture_weights
    mean = (w * mixture_means).sum(-1, keepdim=True)
    return (w * (mixture_stds ** 2 + (mixture_means - mean) ** 2)).sum(-1)


def save_bounds(bounds, output_dir, file_name, n_query_final, T_final):
    """The bounds file of the reference's driver (train_aline.py:271-275): <output_dir>/eval/<stem>_N<n>_T<T>.tar."""
    import os
    path = os.path.join(output_dir, "eval", f"{file_name.split('.')[0]}_N{n_query_final}_T{T_final}.tar")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    torch.save(bounds, path)
    return path


def gather_rows(local, dist=None, world=1):
    """Concatenation over the ranks of per-rank [n_r, ...] tensors with different n_r, in rank order, on every rank
    (one all_gather of the counts, one of the zero-padded blocks)."""
    if dist is None or world <= 1:
        return local
    n = torch.tensor([local.shape[0]], device=local.device, dtype=torch.int64)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    pad = torch.zeros((max(counts),) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    blocks = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(blocks, pad)
    return torch.cat([b[:c] for b, c in zip(blocks, counts)], dim=0)


def bound_statistics(pce, nmc, err_type="se"):
    """Mean and error bar of the bounds over the outer samples (eval.py:176-196)."""
    from .attrdict import AttrDict
    n = pce.shape[0]
    out = {}
    for nm, v in (("pce", pce), ("nmc", nmc)):
        err = torch.std(v, dim=0)
        if err_type == "se":
            err = err / math.sqrt(n)
        elif err_type == "ci":
            err = 1.96 * err / math.sqrt(n)
        elif err_type != "std":
            raise ValueError(f"Unknown err_type: {err_type}")
        out[nm + "_mean"], out[nm + "_err"] = torch.mean(v, dim=0).cpu(), err.cpu()
    return AttrDict(out)


@torch.no_grad()
def eval_boed(model, experiment, T=30, L=int(1e6), M=2000, batch_size=40, time_token=False, stepwise=False,
              err_type="se", dist=None, world=1, rank=0):
    """Final evaluation of the EIG bounds (eval.py:142-198): ceil(M / batch) x (rollout, bounds).  The outer
    batches are independent: with `dist` (torch.distributed, one process per GPU) rank r evaluates batches
    r, r + world, ... and every rank returns the statistics over all of them."""
    model.eval()
    pce_l, nmc_l = [], []
    for i in range((M + batch_size - 1) // batch_size):
        if i % world != rank:
            continue
        theta_0, x, y = get_traces(model, experiment, T, batch_size, time_token)
        pce, nmc = compute_EIG_from_history(experiment, theta_0, x, y, L, batch_size, stepwise)
        pce_l.append(pce)
        nmc_l.append(nmc)
    dev = next(model.parameters()).device
    shape = (0, getattr(experiment, "n_context_init", 1) + T) if stepwise else (0,)   # a rank without batches
    pce = torch.cat(pce_l, dim=0) if pce_l else torch.zeros(shape, device=dev)
    nmc = torch.cat(nmc_l, dim=0) if nmc_l else torch.zeros(shape, device=dev)
    return bound_statistics(gather_rows(pce, dist, world), gather_rows(nmc, dist, world), err_type)

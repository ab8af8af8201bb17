"""Drop-in for `model.base.Aline` (reference model/base.py:11-50): composes the three native
modules; `forward(batch)` runs embedder -> encoder -> head as one C-ABI call
(`aline_step_forward`)."""
import ctypes as C

import torch
import torch.nn as nn

from .. import _lib
from . import _native
from .embedder import Embedder
from .encoder import Encoder
from .head import OutputHead


class _AlineStepFn(torch.autograd.Function):
    """Autograd node of one `Aline.forward(batch)`: HIP forward (aline_step_forward), HIP backward
    (aline_rollout_backward_ex on the step as a T = 1 rollout).  Differentiable outputs: design_out.log_prob
    and posterior_out.mixture_{means,stds,weights} -- what train_aline.py:84-125 back-propagates through."""

    @staticmethod
    def forward(ctx, model, batch, forced_idx, uniform, *params):
        with torch.no_grad():
            out = model._forward_impl(batch, forced_idx, uniform, False)
        ctx.model, ctx.batch = model, {k: _native._get(batch, k) for k in
                                       ("context_x", "context_y", "query_x", "target_x", "target_all", "target_mask")}
        d, p = out.design_out, out.posterior_out
        ctx.idx = d.idx
        ctx.mark_non_differentiable(d.idx, d.zt)
        ctx.lazy_query = out.posterior_out_query
        return d.log_prob, p.mixture_means, p.mixture_stds, p.mixture_weights, d.zt, d.idx

    @staticmethod
    def backward(ctx, g_logp, g_mean, g_std, g_weight, _g_zt, _g_idx):
        from ..train import step_backward
        grads = step_backward(ctx.model, ctx.batch, ctx.idx, g_logp, g_mean, g_std, g_weight)
        return (None, None, None, None, *grads)


class Aline(nn.Module):
    def __init__(self, embedder: Embedder, encoder: Encoder, head: OutputHead,
                 precision: str = None) -> None:
        super().__init__()
        self.embedder, self.encoder, self.head = embedder, encoder, head
        self.range_check = True       # f16x3: read the range status back after every step-API forward (automatic f32 re-run)
        if precision is not None:
            self.set_precision(precision)

    def set_precision(self, precision: str):
        """'f32' / 'f16x3' (reference precision: exact fp32 MFMA / 3-term f16 split), 'bf16' or 'bf16x3'."""
        if precision not in _lib.PREC:
            raise ValueError(precision)
        for mod in (self.embedder, self.encoder, self.head):
            mod.precision = precision
        return self

    @property
    def precision(self):
        return self.encoder.precision

    def model_struct(self):
        m = _lib.AlineModel()
        _native.fill_embedder(m, self.embedder)
        _native.fill_encoder(m, self.encoder)
        _native.fill_head(m, self.head)
        m.precision = _lib.PREC[self.precision]
        return m

    def forward(self, batch, forced_idx=None, uniform=None, return_hidden=False):
        """batch: AttrDict of SURVEY.md 8-b.4.  Extra (optional) arguments are not in the
        reference: `forced_idx` teacher-forces the design (parity tests), `uniform` [B] supplies the
        sampling randoms, `return_hidden` adds `embedding` / `encoding` to the output.

        Under autograd (grad enabled and trainable parameters) the call is one autograd node with a native
        backward, so the reference's own training loop (`loss.backward()`, train_aline.py:128) works on
        the returned `design_out.log_prob` / `posterior_out.*`."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            if return_hidden:
                raise NotImplementedError("aline_amd: return_hidden is forward-only (use torch.no_grad())")
            from ..utils.attrdict import AttrDict
            params = list(self.parameters())
            lp, pm, ps, pw, zt, idx = _AlineStepFn.apply(self, batch, forced_idx, uniform, *params)
            node = lp.grad_fn
            return AttrDict(posterior_out_query=getattr(node, "lazy_query", None) or AttrDict(),
                            posterior_out=AttrDict(mixture_means=pm, mixture_stds=ps, mixture_weights=pw),
                            design_out=AttrDict(idx=idx, log_prob=lp, zt=zt))
        return self._forward_impl(batch, forced_idx, uniform, return_hidden)

    def _forward_impl(self, batch, forced_idx, uniform, return_hidden):
        m = self.model_struct()
        call = _native.StepCall(batch, m.n_theta)
        outs = self.head._prepare(call, batch, forced_idx, uniform)
        z = call.out(call.B, call.N, m.d)
        call.s.encoding = z.data_ptr()
        if return_hidden:
            emb = call.out(call.B, call.N, m.d)
            call.s.embedding = emb.data_ptr()
        ws, nb = call.workspace(m)
        _lib.check(_lib.lib.aline_step_forward(C.byref(m), C.byref(call.s), ws, nb,
                                               _lib.stream_ptr(call.device)), "step_forward")
        if m.precision == _lib.PREC["f16x3"] and self.range_check and not torch.cuda.is_current_stream_capturing():
            # f16 range guard: one 4-byte read-back per step (the reference's own step synchronises too, encoder.py:17)
            rc = _lib.lib.aline_f16_range_status(ws, _lib.stream_ptr(call.device))
            if rc < 0:
                _lib.check(rc, "f16_range_status")
            if rc > 0:
                import warnings
                warnings.warn("aline_amd: an F16X3 operand left f16's range (|x| >= 65504 or non-finite); re-running this step in f32")
                m.precision = _lib.PREC["f32"]
                ws, nb = call.workspace(m)
                _lib.check(_lib.lib.aline_step_forward(C.byref(m), C.byref(call.s), ws, nb,
                                                       _lib.stream_ptr(call.device)), "step_forward")
        frozen = {k: _native._get(batch, k) for k in ("context_x", "query_x", "target_all", "target_mask")}
        out = self.head._package(outs, lambda: self.head._query_posterior(frozen, z))
        if return_hidden:
            out["embedding"], out["encoding"] = emb, z
        return out

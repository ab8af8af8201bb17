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


class Aline(nn.Module):
    def __init__(self, embedder: Embedder, encoder: Encoder, head: OutputHead,
                 precision: str = None) -> None:
        super().__init__()
        self.embedder, self.encoder, self.head = embedder, encoder, head
        if precision is not None:
            self.set_precision(precision)

    def set_precision(self, precision: str):
        """'f32' (reference precision), 'bf16' or 'bf16x3' for the matrix products."""
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
        sampling randoms, `return_hidden` adds `embedding` / `encoding` to the output."""
        _native.require_no_grad(self)
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
        frozen = {k: _native._get(batch, k) for k in ("context_x", "query_x", "target_all", "target_mask")}
        out = self.head._package(outs, lambda: self.head._query_posterior(frozen, z))
        if return_hidden:
            out["embedding"], out["encoding"] = emb, z
        return out

"""Drop-in for `model.encoder.Encoder` (reference model/encoder.py:48-141).  Parameter tree and
state_dict keys equal nn.TransformerEncoder's (`encoder.layers.{l}.self_attn.in_proj_weight`, ...);
forward runs the HIP masked set-attention encoder (C ABI `aline_encoder_forward`)."""
import copy
import ctypes as C

import torch
import torch.nn as nn

from .. import _lib
from . import _native


class _SelfAttnParams(nn.Module):
    """Parameter container with nn.MultiheadAttention's names and init."""

    def __init__(self, d):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * d, d))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * d))
        self.out_proj = nn.Linear(d, d)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.constant_(self.out_proj.bias, 0.0)


class _LayerParams(nn.Module):
    def __init__(self, d, F):
        super().__init__()
        self.self_attn = _SelfAttnParams(d)
        self.linear1 = nn.Linear(d, F)
        self.linear2 = nn.Linear(F, d)
        self.norm1 = nn.LayerNorm(d, eps=1e-5)
        self.norm2 = nn.LayerNorm(d, eps=1e-5)


class _Stack(nn.Module):
    def __init__(self, layer, num_layers):
        super().__init__()
        # nn.TransformerEncoder deep-copies one layer: all layers start identical (encoder.py:79)
        self.layers = nn.ModuleList([copy.deepcopy(layer) for _ in range(num_layers)])


class Encoder(nn.Module):
    def __init__(self, dim_embedding, dim_feedforward, n_head, dropout, num_layers,
                 precision: str = "f32", **kwargs):
        super().__init__()
        if dropout not in (0, 0.0):
            raise ValueError("aline_amd Encoder: dropout must be 0 (config/encoder/encoder.yaml:5)")
        self.dim_embedding, self.dim_feedforward = dim_embedding, dim_feedforward
        self.n_head, self.num_layers, self.precision = n_head, num_layers, precision
        self.encoder = _Stack(_LayerParams(dim_embedding, dim_feedforward), num_layers)

    def forward(self, batch, embeddings):
        """Under autograd the call is one node with a native backward (aline_encoder_backward): gradients wrt the encoder's
        weights and wrt `embeddings`."""
        if _native.wants_grad(self, embeddings):
            return _native.EncoderFn.apply(self, batch, embeddings, *_native.encoder_params(self)[0])
        return self._forward_impl(batch, embeddings)

    def _forward_impl(self, batch, embeddings):
        m = _lib.AlineModel()
        _native.fill_encoder(m, self)
        m.precision = _native.precision_of(self)
        m.embedding_type, m.n_theta = _lib.EMB["data"], 0    # geometry only: n_t rows after queries
        call = _native.StepCall(batch, 0, need_y=False)
        x = _native.f32(embeddings)
        out = call.out(call.B, call.N, self.dim_embedding)
        call.s.encoding = out.data_ptr()
        ws, nb = call.workspace(m)
        _lib.check(_lib.lib.aline_encoder_forward(C.byref(m), C.byref(call.s), x.data_ptr(), ws, nb,
                                                  _lib.stream_ptr(call.device)), "encoder_forward")
        return out

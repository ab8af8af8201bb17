"""Drop-in for `model.embedder.Embedder` (reference model/embedder.py:6-214): same constructor
kwargs, same state_dict keys, forward runs the HIP embedder (C ABI `aline_embed_forward`)."""
import ctypes as C
from typing import Any

import torch
import torch.nn as nn

from .. import _lib
from . import _native


class Embedder(nn.Module):
    def __init__(self, dim_x: int, dim_y: int, dim_embedding: int, dim_feedforward: int,
                 n_target_theta: int = 0, embedding_type: str = "data", precision: str = "f32",
                 **kwargs: Any) -> None:
        super().__init__()
        self.dim_x, self.dim_y, self.dim_embedding = dim_x, dim_y, dim_embedding
        self.n_target_theta, self.embedding_type = n_target_theta, embedding_type
        self.precision = precision
        # parameter containers only (keys x_embedder.{0,2}.*); forward never calls them
        self.x_embedder = nn.Sequential(nn.Linear(dim_x, dim_feedforward), nn.ReLU(),
                                        nn.Linear(dim_feedforward, dim_embedding))
        self.y_embedder = nn.Sequential(nn.Linear(dim_y, dim_feedforward), nn.ReLU(),
                                        nn.Linear(dim_feedforward, dim_embedding))
        if embedding_type not in ("data", "theta", "mix"):
            raise ValueError(f"Unknown embedding type: {embedding_type}")   # embedder.py:93
        if embedding_type in ("theta", "mix"):
            if n_target_theta <= 0:                                       # embedder.py:61-62
                raise ValueError("dim_theta must be positive for theta or mix embedding type")
            self.theta_tokens = nn.Parameter(torch.randn(n_target_theta, dim_embedding))

    def forward(self, batch) -> torch.Tensor:
        """Under autograd (trainable parameters) the call is one node with a native backward (aline_embed_backward), so the
        reference's own Aline composition (model/base.py:47-50) trains with this module swapped in by `_target_` alone."""
        if _native.wants_grad(self):
            return _native.EmbedFn.apply(self, batch, *_native.embedder_params(self)[0])
        return self._forward_impl(batch)

    def _forward_impl(self, batch) -> torch.Tensor:
        m = _lib.AlineModel()
        _native.fill_embedder(m, self)
        m.precision = _native.precision_of(self)
        call = _native.StepCall(batch, m.n_theta)
        out = call.out(call.B, call.N, self.dim_embedding)
        call.s.embedding = out.data_ptr()
        ws, nb = call.workspace(m)
        _lib.check(_lib.lib.aline_embed_forward(C.byref(m), C.byref(call.s), ws, nb,
                                                _lib.stream_ptr(call.device)), "embed_forward")
        return out

"""Glue between the nn.Module mirrors and the C ABI: fills `aline_model` / `aline_step` from
module parameters and batch tensors.  No arithmetic happens in Python."""
import ctypes as C

import torch

from .. import _lib
from .._lib import AlineModel, AlineStep, f32, ptr

_ws = _lib.Workspace()


def _get(batch, key):
    if isinstance(batch, dict):
        return batch.get(key)
    return getattr(batch, key, None)


def require_no_grad(module):
    if torch.is_grad_enabled() and any(p.requires_grad for p in module.parameters()):
        raise NotImplementedError(
            "aline_amd: the HIP path of this build is forward-only; call under torch.no_grad() "
            "(backward kernels are not built yet)")


def fill_embedder(m: AlineModel, emb):
    m.dim_x, m.dim_y, m.d = emb.dim_x, emb.dim_y, emb.dim_embedding
    m.F = emb.x_embedder[0].out_features
    m.embedding_type = _lib.EMB[emb.embedding_type]
    m.n_theta = emb.n_target_theta if emb.embedding_type in ("theta", "mix") else 0
    for nm in ("x", "y"):
        seq = getattr(emb, f"{nm}_embedder")
        setattr(m, f"{nm}_w1", seq[0].weight.data_ptr())
        setattr(m, f"{nm}_b1", seq[0].bias.data_ptr())
        setattr(m, f"{nm}_w2", seq[2].weight.data_ptr())
        setattr(m, f"{nm}_b2", seq[2].bias.data_ptr())
    m.theta_tokens = emb.theta_tokens.data_ptr() if m.n_theta > 0 else None


def fill_encoder(m: AlineModel, enc):
    m.d, m.F, m.H, m.L = enc.dim_embedding, enc.dim_feedforward, enc.n_head, enc.num_layers
    for l, layer in enumerate(enc.encoder.layers):
        m.in_proj_w[l] = layer.self_attn.in_proj_weight.data_ptr()
        m.in_proj_b[l] = layer.self_attn.in_proj_bias.data_ptr()
        m.out_proj_w[l] = layer.self_attn.out_proj.weight.data_ptr()
        m.out_proj_b[l] = layer.self_attn.out_proj.bias.data_ptr()
        m.lin1_w[l], m.lin1_b[l] = layer.linear1.weight.data_ptr(), layer.linear1.bias.data_ptr()
        m.lin2_w[l], m.lin2_b[l] = layer.linear2.weight.data_ptr(), layer.linear2.bias.data_ptr()
        m.norm1_w[l], m.norm1_b[l] = layer.norm1.weight.data_ptr(), layer.norm1.bias.data_ptr()
        m.norm2_w[l], m.norm2_b[l] = layer.norm2.weight.data_ptr(), layer.norm2.bias.data_ptr()


def fill_head(m: AlineModel, head):
    th = head.target_head
    m.d, m.F, m.C = th.dim_embedding, th.dim_feedforward, th.num_components
    m.dim_y = head.dim_y
    m.std_min = th.std_min
    m.time_token = 1 if head.time_token else 0
    pr = head.acquisition_head.predictor
    m.acq_w1, m.acq_b1 = pr[0].weight.data_ptr(), pr[0].bias.data_ptr()
    m.acq_w2, m.acq_b2 = pr[2].weight.data_ptr(), pr[2].bias.data_ptr()
    for c, h in enumerate(th.heads):
        m.gmm_w1[c], m.gmm_b1[c] = h[0].weight.data_ptr(), h[0].bias.data_ptr()
        m.gmm_w2[c], m.gmm_b2[c] = h[2].weight.data_ptr(), h[2].bias.data_ptr()


class StepCall:
    """Holds the tensors of one call alive and exposes the filled `aline_step`."""

    def __init__(self, batch, n_theta, need_y=True):
        self.keep = []
        s = AlineStep()
        cx = self._in(_get(batch, "context_x"))
        qx = self._in(_get(batch, "query_x"))
        self.device = cx.device
        s.B, s.n_ctx, s.n_query = cx.shape[0], cx.shape[1], qx.shape[1]
        s.context_x, s.query_x = ptr(cx), ptr(qx)
        if need_y:
            s.context_y = ptr(self._in(_get(batch, "context_y")))
        ta = _get(batch, "target_all")
        n_t = ta.shape[1]
        self.n_t = n_t
        s.n_target_data = n_t - n_theta
        tx = _get(batch, "target_x")
        if tx is not None and s.n_target_data > 0:
            s.target_x = ptr(self._in(tx))
        s.target_all = ptr(self._in(ta.reshape(ta.shape[0], n_t)))
        tm = _get(batch, "target_mask")
        if tm is not None:
            s.target_mask = ptr(self._keep(tm.to(device=self.device, dtype=torch.uint8).contiguous()))
        self.s = s
        self.B, self.n_ctx, self.n_query = s.B, s.n_ctx, s.n_query
        self.N = s.n_ctx + s.n_query + n_t

    def _keep(self, t):
        self.keep.append(t)
        return t

    def _in(self, t):
        if not t.is_cuda:
            raise RuntimeError("aline_amd: batch tensors must live on the GPU (no CPU fallback)")
        return self._keep(f32(t))

    def out(self, *shape, dtype=torch.float32):
        return self._keep(torch.empty(*shape, dtype=dtype, device=self.device))

    def workspace(self, model_struct):
        nbytes = _lib.lib.aline_step_workspace_bytes(C.byref(model_struct), C.byref(self.s))
        if nbytes == 0:
            raise RuntimeError("aline_amd: unsupported model/batch configuration")
        buf = _ws.get(nbytes, self.device)
        return buf.data_ptr(), buf.numel()


def precision_of(module):
    return _lib.PREC[getattr(module, "precision", "f32")]

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


def wants_grad(module, *tensors):
    return torch.is_grad_enabled() and (any(p.requires_grad for p in module.parameters())
                                        or any(torch.is_tensor(t) and t.requires_grad for t in tensors))


_bwd_ws = _lib.Workspace()


class StageStep:
    """One step of a stand-alone stage as the T = 1 rollout the stage backward entry points take (include/aline_hip.h):
    role = 1..n_ctx on the context points, 0 on the queries; slot = chosen point."""

    def __init__(self, batch, n_theta, idx=None):
        cx, qx = f32(_get(batch, "context_x")), f32(_get(batch, "query_x"))
        dev = self.device = cx.device
        B, n_c, n_q = cx.shape[0], cx.shape[1], qx.shape[1]
        self.B, self.P = B, n_c + n_q
        ta = _get(batch, "target_all")
        n_t = ta.shape[1]
        self.N = self.P + n_t
        keep = self.keep = []
        px = torch.cat([cx, qx], dim=1).contiguous()
        cy = _get(batch, "context_y")
        dy = cy.shape[-1] if cy is not None else 1
        py = torch.zeros(B, self.P, dy, device=dev)
        if cy is not None:
            py[:, :n_c] = f32(cy)
        role = torch.zeros(B, self.P, dtype=torch.int32, device=dev)
        role[:, :n_c] = torch.arange(1, n_c + 1, dtype=torch.int32, device=dev)
        target_all = f32(ta.reshape(B, n_t))
        n_td = n_t - n_theta
        tx = _get(batch, "target_x")
        tx = f32(tx) if (tx is not None and n_td > 0) else None
        tm = _get(batch, "target_mask")
        tmask = None if tm is None else tm.to(dev, torch.uint8).contiguous()
        slot = None if idx is None else (n_c + idx.reshape(B).to(torch.int32)).contiguous()
        keep += [px, py, role, target_all, tx, tmask, slot]
        r = self.r = _lib.AlineRollout()
        r.B, r.P, r.n_ctx0, r.n_target_data, r.T = B, self.P, n_c, n_td, 1
        r.point_x, r.point_y, r.role = px.data_ptr(), py.data_ptr(), role.data_ptr()
        r.target_x, r.target_all, r.target_mask = ptr(tx), target_all.data_ptr(), ptr(tmask)
        r.slot = ptr(slot)

    def workspace(self, m):
        nbytes = _lib.lib.aline_rollout_backward_workspace_bytes(C.byref(m), C.byref(self.r), 1)
        if nbytes == 0:
            raise RuntimeError("aline_amd: unsupported configuration for backward")
        buf = _bwd_ws.get(nbytes, self.device)
        return buf.data_ptr(), buf.numel()


def grads_struct(params, names):
    """aline_grads whose members `names` (in order) point at fresh zero tensors shaped like `params`; returns (struct, tensors)."""
    g = _lib.AlineGrads()
    outs = []
    for p, nm in zip(params, names):
        t = torch.zeros_like(p)
        outs.append(t)
        if isinstance(nm, tuple):
            getattr(g, nm[0])[nm[1]] = t.data_ptr()
        else:
            setattr(g, nm, t.data_ptr())
    return g, outs


def embedder_params(emb):
    ps, nm = [], []
    for k in ("x", "y"):
        seq = getattr(emb, f"{k}_embedder")
        ps += [seq[0].weight, seq[0].bias, seq[2].weight, seq[2].bias]
        nm += [f"{k}_w1", f"{k}_b1", f"{k}_w2", f"{k}_b2"]
    if getattr(emb, "theta_tokens", None) is not None and emb.embedding_type in ("theta", "mix"):
        ps.append(emb.theta_tokens)
        nm.append("theta_tokens")
    return ps, nm


def encoder_params(enc):
    ps, nm = [], []
    for l, layer in enumerate(enc.encoder.layers):
        ps += [layer.self_attn.in_proj_weight, layer.self_attn.in_proj_bias, layer.self_attn.out_proj.weight,
               layer.self_attn.out_proj.bias, layer.linear1.weight, layer.linear1.bias, layer.linear2.weight,
               layer.linear2.bias, layer.norm1.weight, layer.norm1.bias, layer.norm2.weight, layer.norm2.bias]
        nm += [(k, l) for k in ("in_proj_w", "in_proj_b", "out_proj_w", "out_proj_b", "lin1_w", "lin1_b", "lin2_w",
                                "lin2_b", "norm1_w", "norm1_b", "norm2_w", "norm2_b")]
    return ps, nm


def head_params(head):
    pr = head.acquisition_head.predictor
    ps = [pr[0].weight, pr[0].bias, pr[2].weight, pr[2].bias]
    nm = ["acq_w1", "acq_b1", "acq_w2", "acq_b2"]
    for c, h in enumerate(head.target_head.heads):
        ps += [h[0].weight, h[0].bias, h[2].weight, h[2].bias]
        nm += [("gmm_w1", c), ("gmm_b1", c), ("gmm_w2", c), ("gmm_b2", c)]
    return ps, nm


class EmbedFn(torch.autograd.Function):
    """Embedder.forward as an autograd node: HIP forward (aline_embed_forward), HIP backward (aline_embed_backward)."""

    @staticmethod
    def forward(ctx, module, batch, *params):
        with torch.no_grad():
            out = module._forward_impl(batch)
        ctx.module = module
        ctx.batch = {k: _get(batch, k) for k in ("context_x", "context_y", "query_x", "target_x", "target_all", "target_mask")}
        return out

    @staticmethod
    def backward(ctx, dx):
        mod = ctx.module
        m = AlineModel()
        fill_embedder(m, mod)
        m.precision = _lib.PREC["f32"]
        st = StageStep(ctx.batch, m.n_theta)
        ps, nm = embedder_params(mod)
        g, outs = grads_struct(ps, nm)
        ws, nb = st.workspace(m)
        _lib.check(_lib.lib.aline_embed_backward(C.byref(m), C.byref(st.r), f32(dx).data_ptr(), C.byref(g), ws, nb,
                                                 _lib.stream_ptr(st.device)), "embed_backward")
        return (None, None, *outs)


class EncoderFn(torch.autograd.Function):
    """Encoder.forward as an autograd node (aline_encoder_forward / aline_encoder_backward): gradients wrt the encoder's
    weights AND wrt the embeddings it was given."""

    @staticmethod
    def forward(ctx, module, batch, embeddings, *params):
        with torch.no_grad():
            out = module._forward_impl(batch, embeddings)
        ctx.module, ctx.x = module, f32(embeddings)
        ctx.batch = {k: _get(batch, k) for k in ("context_x", "context_y", "query_x", "target_x", "target_all", "target_mask")}
        return out

    @staticmethod
    def backward(ctx, dz):
        mod = ctx.module
        m = AlineModel()
        fill_encoder(m, mod)
        m.precision = _lib.PREC["f32"]
        m.embedding_type, m.n_theta = _lib.EMB["data"], 0    # geometry only: n_t rows after the queries
        st = StageStep(ctx.batch, 0)
        ps, nm = encoder_params(mod)
        g, outs = grads_struct(ps, nm)
        dx = torch.empty_like(ctx.x)
        ws, nb = st.workspace(m)
        _lib.check(_lib.lib.aline_encoder_backward(C.byref(m), C.byref(st.r), ctx.x.data_ptr(), f32(dz).data_ptr(), C.byref(g),
                                                   dx.data_ptr(), ws, nb, _lib.stream_ptr(st.device)), "encoder_backward")
        return (None, None, dx, *outs)


class HeadFn(torch.autograd.Function):
    """OutputHead.forward as an autograd node (aline_head_forward / aline_head_backward).  Differentiable outputs:
    design_out.log_prob and posterior_out.mixture_{means,stds,weights}; gradients wrt the head's weights and wrt z."""

    @staticmethod
    def forward(ctx, module, batch, z, forced_idx, uniform, *params):
        with torch.no_grad():
            out = module._forward_impl(batch, z, forced_idx, uniform)
        ctx.module, ctx.z = module, f32(z)
        ctx.batch = {k: _get(batch, k) for k in ("context_x", "context_y", "query_x", "target_x", "target_all", "target_mask")}
        d, p = out.design_out, out.posterior_out
        ctx.idx = d.idx
        ctx.mark_non_differentiable(d.idx, d.zt)
        ctx.lazy_query = out.posterior_out_query
        return d.log_prob, p.mixture_means, p.mixture_stds, p.mixture_weights, d.zt, d.idx

    @staticmethod
    def backward(ctx, g_logp, g_mean, g_std, g_weight, _g_zt, _g_idx):
        mod = ctx.module
        m = AlineModel()
        fill_head(m, mod)
        m.precision = _lib.PREC["f32"]
        B, N, d = ctx.z.shape
        n_t = _get(ctx.batch, "target_all").shape[1]
        m.embedding_type, m.n_theta = _lib.EMB["data"], 0    # geometry only: n_t rows after the queries
        st = StageStep(ctx.batch, 0, ctx.idx)
        ps, nm = head_params(mod)
        g, outs = grads_struct(ps, nm)
        dev = ctx.z.device
        glp = f32(g_logp.reshape(B, 1)) if g_logp is not None else torch.zeros(B, 1, device=dev)
        gm, gs, gw = (None if t is None else f32(t.reshape(1, B, n_t, -1)) for t in (g_mean, g_std, g_weight))
        if gm is None and gs is None and gw is None:
            gm = torch.zeros(1, B, n_t, m.C, device=dev)
        dz = torch.empty_like(ctx.z)
        ws, nb = st.workspace(m)
        _lib.check(_lib.lib.aline_head_backward(C.byref(m), C.byref(st.r), ctx.z.data_ptr(), glp.data_ptr(), ptr(gm), ptr(gs),
                                                ptr(gw), C.byref(g), dz.data_ptr(), ws, nb, _lib.stream_ptr(dev)),
                   "head_backward")
        return (None, None, dz, None, None, *outs)


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

"""Drop-in for `model.head.OutputHead` (reference model/head.py:269-393) with its
`AcquisitionHead` (:9-44) and `GMMTargetHead` (:115-266) parameter trees; forward runs the HIP
head (C ABI `aline_head_forward`)."""
import ctypes as C
from typing import Any

import torch
import torch.nn as nn

from .. import _lib
from ..utils.attrdict import AttrDict
from . import _native


class AcquisitionHead(nn.Module):
    def __init__(self, dim_embedding: int, dim_feedforward: int, time_token: bool, **kwargs: Any):
        super().__init__()
        if time_token:
            dim_embedding += 1
        self.predictor = nn.Sequential(nn.Linear(dim_embedding, dim_feedforward), nn.ReLU(),
                                       nn.Linear(dim_feedforward, 1), nn.Flatten(start_dim=-2),
                                       nn.Softmax(dim=-1))


class GMMTargetHead(nn.Module):
    def __init__(self, dim_y: int, dim_embedding: int, dim_feedforward: int, num_components: int,
                 single_head: bool = False, std_min: float = 1e-4, **kwargs: Any):
        super().__init__()
        if single_head:
            raise NotImplementedError("aline_amd: single_head=True is not used by any reference config")
        if dim_y != 1:
            raise NotImplementedError("aline_amd: GMM head supports dim_y == 1 (as the reference, head.py:113)")
        self.dim_embedding, self.dim_feedforward, self.dim_y = dim_embedding, dim_feedforward, dim_y
        self.num_components, self.std_min = num_components, std_min
        self.heads = nn.ModuleList([
            nn.Sequential(nn.Linear(dim_embedding, dim_feedforward), nn.ReLU(),
                          nn.Linear(dim_feedforward, dim_y * 3)) for _ in range(num_components)])

    @staticmethod
    def compute_ll(value, means, stds, weights):
        from ..utils.eval import compute_ll
        return compute_ll(value, means, stds, weights)


class LazyQueryPosterior(AttrDict):
    """`posterior_out_query` (head.py:366) is half of the forward FLOPs and unused by train/eval;
    it is computed on first access of any of its three fields."""

    def __init__(self, compute):
        super().__init__()
        object.__setattr__(self, "_compute", compute)

    def _materialise(self):
        if "mixture_means" not in self:
            m, s, w = object.__getattribute__(self, "_compute")()
            dict.__setitem__(self, "mixture_means", m)
            dict.__setitem__(self, "mixture_stds", s)
            dict.__setitem__(self, "mixture_weights", w)

    def __getattr__(self, k):
        if k in ("mixture_means", "mixture_stds", "mixture_weights"):
            self._materialise()
            return dict.__getitem__(self, k)
        raise AttributeError(k)

    def __getitem__(self, k):
        if k in ("mixture_means", "mixture_stds", "mixture_weights"):
            self._materialise()
        return dict.__getitem__(self, k)


class OutputHead(nn.Module):
    def __init__(self, dim_x: int, dim_y: int, dim_embedding: int, dim_feedforward: int,
                 num_components: int = 10, single_head: bool = False, std_min: float = 1e-4,
                 value_head: bool = False, time_token: bool = False, precision: str = "f32",
                 lazy_query_posterior: bool = True, **kwargs: Any) -> None:
        super().__init__()
        if value_head:
            raise NotImplementedError("aline_amd: value_head is not instantiated by any reference config")
        self.dim_x, self.dim_y, self.time_token = dim_x, dim_y, time_token
        self.precision, self.lazy_query_posterior = precision, lazy_query_posterior
        self.acquisition_head = AcquisitionHead(dim_embedding=dim_embedding,
                                                dim_feedforward=dim_feedforward, time_token=time_token)
        self.target_head = GMMTargetHead(dim_y=dim_y, dim_embedding=dim_embedding,
                                         dim_feedforward=dim_feedforward,
                                         num_components=num_components, single_head=single_head,
                                         std_min=std_min)
        self.value_head = False

    # -- shared by OutputHead.forward and Aline.forward -------------------------------------------
    def _prepare(self, call, batch, forced_idx=None, uniform=None, with_query=None):
        s, C_ = call.s, self.target_head.num_components
        B, nq, n_t = call.B, call.n_query, call.n_t
        outs = dict(idx=call.out(B, 1, dtype=torch.int64), log_prob=call.out(B), zt=call.out(B, nq),
                    pm=call.out(B, n_t, C_), ps=call.out(B, n_t, C_), pw=call.out(B, n_t, C_))
        s.idx, s.log_prob, s.zt = outs["idx"].data_ptr(), outs["log_prob"].data_ptr(), outs["zt"].data_ptr()
        s.post_mean, s.post_std, s.post_weight = (outs["pm"].data_ptr(), outs["ps"].data_ptr(),
                                                  outs["pw"].data_ptr())
        if self.time_token:
            t = _native._get(batch, "t")
            s.time_t = call._in(t.reshape(-1)[:1].to(call.device)).data_ptr()
        if forced_idx is not None:
            s.select_mode = _lib.SELECT_FORCED
            s.forced_idx = call._keep(forced_idx.reshape(-1).to(call.device, torch.int64).contiguous()).data_ptr()
        elif self.training:
            s.select_mode = _lib.SELECT_SAMPLE                       # head.py:350-354
            u = uniform if uniform is not None else torch.rand(B, device=call.device)
            s.uniform = call._keep(u.to(call.device, torch.float32).contiguous()).data_ptr()
        else:
            s.select_mode = _lib.SELECT_ARGMAX                       # head.py:355-358
        if with_query is None:
            with_query = not self.lazy_query_posterior
        if with_query:
            outs["qm"], outs["qs"], outs["qw"] = (call.out(B, nq, C_), call.out(B, nq, C_),
                                                  call.out(B, nq, C_))
            s.postq_mean, s.postq_std, s.postq_weight = (outs["qm"].data_ptr(), outs["qs"].data_ptr(),
                                                         outs["qw"].data_ptr())
        return outs

    def _package(self, outs, lazy_compute):
        if "qm" in outs:
            pq = AttrDict(mixture_means=outs["qm"], mixture_stds=outs["qs"], mixture_weights=outs["qw"])
        else:
            pq = LazyQueryPosterior(lazy_compute)
        return AttrDict(
            posterior_out_query=pq,
            posterior_out=AttrDict(mixture_means=outs["pm"], mixture_stds=outs["ps"],
                                   mixture_weights=outs["pw"]),
            design_out=AttrDict(idx=outs["idx"], log_prob=outs["log_prob"], zt=outs["zt"]))

    def _query_posterior(self, batch, z):
        """posterior_out_query only (head.py:366), run on demand."""
        m = _lib.AlineModel()
        _native.fill_head(m, self)
        m.precision = _native.precision_of(self)
        m.embedding_type, m.n_theta = _lib.EMB["data"], 0
        call = _native.StepCall(batch, 0, need_y=False)
        C_ = self.target_head.num_components
        q = [call.out(call.B, call.n_query, C_) for _ in range(3)]
        call.s.postq_mean, call.s.postq_std, call.s.postq_weight = (t.data_ptr() for t in q)
        call.s.select_mode = _lib.SELECT_ARGMAX
        ws, nb = call.workspace(m)
        _lib.check(_lib.lib.aline_head_forward(C.byref(m), C.byref(call.s), z.data_ptr(), ws, nb,
                                               _lib.stream_ptr(call.device)), "head_forward")
        return tuple(q)

    def forward(self, batch, z, forced_idx=None, uniform=None):
        """Under autograd the call is one node with a native backward (aline_head_backward): `design_out.log_prob` and
        `posterior_out.mixture_*` are differentiable wrt the head's weights and wrt `z`."""
        if _native.wants_grad(self, z):
            from ..utils.attrdict import AttrDict as _AD
            lp, pm, ps, pw, zt, idx = _native.HeadFn.apply(self, batch, z, forced_idx, uniform, *_native.head_params(self)[0])
            node = lp.grad_fn
            return _AD(posterior_out_query=getattr(node, "lazy_query", None) or _AD(),
                       posterior_out=_AD(mixture_means=pm, mixture_stds=ps, mixture_weights=pw),
                       design_out=_AD(idx=idx, log_prob=lp, zt=zt))
        return self._forward_impl(batch, z, forced_idx, uniform)

    def _forward_impl(self, batch, z, forced_idx=None, uniform=None):
        m = _lib.AlineModel()
        _native.fill_head(m, self)
        m.precision = _native.precision_of(self)
        m.embedding_type, m.n_theta = _lib.EMB["data"], 0
        call = _native.StepCall(batch, 0, need_y=False)
        z = _native.f32(z)
        outs = self._prepare(call, batch, forced_idx, uniform)
        ws, nb = call.workspace(m)
        _lib.check(_lib.lib.aline_head_forward(C.byref(m), C.byref(call.s), z.data_ptr(), ws, nb,
                                               _lib.stream_ptr(call.device)), "head_forward")
        frozen = {k: _native._get(batch, k) for k in ("context_x", "query_x", "target_all", "target_mask")}
        return self._package(outs, lambda: self._query_posterior(frozen, z))

"""Mirror of the reference's `HiddenLocation` simulator (tasks/location_finding.py:8-192): the
input generator of the north-star config.  Sampling is plain device-side torch (plumbing, already
vectorised in the reference); the likelihood used by the EIG bounds is the HIP kernel."""
import math

import torch

from .. import _lib
from ..utils.attrdict import AttrDict
from .base_task import Task


class HiddenLocation(Task):
    def __init__(self, name: str = "Location", dim_x: int = 2, dim_y: int = 1, embedding_type="theta",
                 n_target_theta: int = 2, n_context_init: int = 1, n_query_init: int = 200, K: int = 1,
                 theta_dist="uniform", design_scale=None, outcome_scale=10, noise_scale=0.5,
                 base_signal: float = 0.1, max_signal: float = 1e-4, device=None, **kwargs) -> None:
        super().__init__(dim_x=dim_x, dim_y=dim_y, device=device)
        if theta_dist != "uniform":
            raise NotImplementedError("aline_amd HiddenLocation: only the uniform prior of "
                                      "config/task/location_finding.yaml is built")
        assert n_target_theta == K * dim_x, "n_theta must be equal to K * dim_x"
        self.name, self.K = name, K
        self.design_scale = 1.0 if design_scale is None else float(design_scale)
        self.noise_scale, self.base_signal, self.max_signal = float(noise_scale), base_signal, max_signal
        self.n_target_theta, self.n_context_init, self.n_query_init = n_target_theta, n_context_init, n_query_init
        self.embedding_type = embedding_type

    @torch.no_grad()
    def sample_theta(self, batch_size):
        shape = [batch_size] if isinstance(batch_size, int) else list(batch_size)
        return torch.rand(*shape, self.K, self.dim_x, device=self.device)      # U[0,1]^{K x D}

    @torch.no_grad()
    def sample_data(self, batch_size, n_data):
        return torch.rand(batch_size, n_data, self.dim_x, device=self.device)

    def total_density(self, xi, theta):
        sq = (xi.unsqueeze(-2).expand(theta.shape) - theta).pow(2).sum(-1)
        return torch.log(self.base_signal + (self.max_signal + sq).pow(-1).sum(-1, keepdim=True))

    def forward(self, xi, theta):
        signal = self.total_density(xi, theta)
        return signal + self.noise_scale * torch.randn_like(signal)

    def log_likelihood(self, y, xi, theta):
        """[L, B, K, D] thetas against y [1, B, 1], xi [1, B, D] -> [L, B, 1] (location_finding.py:149-164)."""
        L1, B = theta.shape[0], theta.shape[1]
        S = torch.zeros(L1, B, device=theta.device)
        self.native_eig_step(_lib.f32(theta), _lib.f32(xi).reshape(B, -1), _lib.f32(y).reshape(B), S)
        return S.unsqueeze(-1)

    def native_eig_step(self, thetas, xi, y, S):
        L1, B = S.shape
        _lib.check(_lib.lib.aline_eig_location_step(
            thetas.data_ptr(), xi.reshape(B, -1).contiguous().data_ptr(),
            y.reshape(B).contiguous().data_ptr(), S.data_ptr(), L1, B, self.K, self.dim_x,
            self.noise_scale, self.base_signal, self.max_signal, _lib.stream_ptr(S.device)),
            "eig_location_step")

    _hist_ws = _lib.Workspace()

    def native_eig_history(self, thetas, x, y):
        """Stepwise sPCE / sNMC bounds of a whole design history in ONE pass over the contrastive samples (C ABI
        aline_eig_location_history): thetas [L + 1, B, K, D] with row 0 the true parameter, x [B, T, D] unnormalised designs in
        order of acquisition, y [B, T(, 1)] -> (pce [B, T], nmc [B, T]) = what utils/eval.py:64-78 returns with stepwise=True."""
        L1, B = thetas.shape[0], thetas.shape[1]
        T = x.shape[1]
        th, xx, yy = _lib.f32(thetas), _lib.f32(x).reshape(B, T, -1).contiguous(), _lib.f32(y).reshape(B, T).contiguous()
        pce, nmc = torch.empty(B, T, device=th.device), torch.empty(B, T, device=th.device)
        nb = _lib.lib.aline_eig_history_workspace_bytes(L1, B, T)
        ws = self._hist_ws.get(nb, th.device)
        _lib.check(_lib.lib.aline_eig_location_history(th.data_ptr(), xx.data_ptr(), yy.data_ptr(), L1, B, T, self.K, self.dim_x,
                                                       self.noise_scale, self.base_signal, self.max_signal, pce.data_ptr(), nmc.data_ptr(),
                                                       ws.data_ptr(), ws.numel(), _lib.stream_ptr(th.device)), "eig_location_history")
        return pce, nmc

    @torch.no_grad()
    def sample_batch(self, batch_size, with_query=True):
        theta = self.sample_theta(batch_size)
        if not with_query:
            self.n_query_init = 1
        n = self.n_context_init + self.n_query_init
        x = self.sample_data(batch_size, n)
        y = self.forward(self.unnormalise_design(x),
                         theta.unsqueeze(1).expand(batch_size, n, self.K, self.dim_x))
        theta = theta.reshape(batch_size, self.n_target_theta, 1)
        batch = AttrDict()
        batch.context_x = x[:, :self.n_context_init].contiguous()
        batch.context_y = y[:, :self.n_context_init].contiguous()
        batch.query_x = x[:, self.n_context_init:].contiguous()
        batch.query_y = y[:, self.n_context_init:].contiguous()
        batch.target_all = batch.target_theta = theta
        batch.n_target_theta = self.n_target_theta
        return batch

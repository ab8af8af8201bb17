"""Mirror of the reference's `CESTask` (tasks/ces.py:9-247): constant-elasticity-of-substitution
preference simulator.  Sampling is device-side torch (input generator); the likelihood used by the
EIG bounds (ces.py:169-210 + CensoredSigmoidNormal.log_prob) is the HIP kernel."""
import torch

from .. import _lib
from ..utils.attrdict import AttrDict
from .base_task import Task


class CESTask(Task):
    def __init__(self, name: str = "CES", dim_x: int = 6, dim_y: int = 1, embedding_type="theta",
                 n_theta: int = 5, n_context_init: int = 5, n_query_init: int = 300,
                 design_scale: int = 100, noise_scale: float = 0.005, epsilon: float = 2 ** (-22),
                 device=None, **kwargs) -> None:
        super().__init__(dim_x=dim_x, dim_y=dim_y, device=device)
        self.name, self.basket_dim, self.n_theta = name, 3, n_theta
        self.n_target_theta = n_theta
        self.n_context_init, self.n_query_init = n_context_init, n_query_init
        self.design_scale, self.noise_scale, self.epsilon = design_scale, noise_scale, epsilon
        self.embedding_type = embedding_type
        self.u_mu, self.u_sigma = 1.0, 3.0

    @torch.no_grad()
    def sample_theta(self, batch_size):
        shape = [batch_size] if isinstance(batch_size, int) else list(batch_size)
        dev = self.device
        rho = 0.01 + 0.99 * torch.rand(*shape, device=dev)                       # Beta(1,1), ces.py:63-66
        g = -torch.log(torch.rand(*shape, 3, device=dev).clamp_min(1e-30))       # Dirichlet(1,1,1)
        alpha = g / g.sum(-1, keepdim=True)
        log_u = self.u_mu + self.u_sigma * torch.randn(*shape, device=dev)
        return torch.cat([rho.unsqueeze(-1), alpha, log_u.unsqueeze(-1)], dim=-1)  # [.., 5]

    @torch.no_grad()
    def sample_data(self, batch_size, n_data):
        return torch.rand(batch_size, n_data, 6, device=self.device) * self.design_scale

    def normalise_design(self, x):
        return x

    def unnormalise_design(self, x):
        return x

    def utility(self, x, rho, alpha):
        return torch.sum(alpha * x ** rho, dim=-1, keepdim=True) ** (1.0 / rho)

    def forward(self, xi, theta):
        rho, alpha, u = theta[..., 0:1], theta[..., 1:4], torch.exp(theta[..., 4:5])
        xi = torch.clamp(xi, min=0.01, max=100.0)
        b1, b2 = xi[..., :3], xi[..., 3:]
        mu = (self.utility(b1, rho, alpha) - self.utility(b2, rho, alpha)) * u
        sigma = (1 + torch.norm(b1 - b2, dim=-1, p=2, keepdim=True)) * self.noise_scale * u
        y = torch.sigmoid(mu + sigma * torch.randn_like(mu))
        return torch.clamp(y, min=self.epsilon, max=1 - self.epsilon)           # censoring, csn.py:42-45

    def log_likelihood(self, y, xi, theta):
        """theta [L, B, 5] against y [1, B, 1], xi [1, B, 6] -> [L, B, 1] (ces.py:169-210)."""
        L1, B = theta.shape[0], theta.shape[1]
        S = torch.zeros(L1, B, device=theta.device)
        self.native_eig_step(_lib.f32(theta), _lib.f32(xi).reshape(B, 6), _lib.f32(y).reshape(B), S)
        return S.unsqueeze(-1)

    def native_eig_step(self, thetas, xi, y, S, nan_flag=None):
        L1, B = S.shape
        _lib.check(_lib.lib.aline_eig_ces_step(
            thetas.contiguous().data_ptr(), xi.reshape(B, 6).contiguous().data_ptr(),
            y.reshape(B).contiguous().data_ptr(), S.data_ptr(), L1, B, self.noise_scale, self.epsilon,
            None if nan_flag is None else nan_flag.data_ptr(), _lib.stream_ptr(S.device)), "eig_ces_step")

    _hist_ws = _lib.Workspace()

    def native_eig_history(self, thetas, x, y):
        """Stepwise sPCE / sNMC bounds of a whole design history in ONE pass over the contrastive samples (C ABI aline_eig_ces_history):
        thetas [L + 1, B, 5] with row 0 the true parameter, x [B, T, 6] designs in order of acquisition, y [B, T(, 1)] ->
        (pce [B, T], nmc [B, T]) = what utils/eval.py:64-78 returns with stepwise=True.  Returns None where the kernel does not apply
        (more than 16 steps, or a per-(episode, step) table beyond 64 KB): the caller keeps the step kernels."""
        L1, B = thetas.shape[0], thetas.shape[1]
        T = x.shape[1]
        if T > 16 or B * T * 96 > 64 * 1024:
            return None
        th = _lib.f32(thetas).reshape(L1, B, 5).contiguous()
        xx, yy = _lib.f32(x).reshape(B, T, 6).contiguous(), _lib.f32(y).reshape(B, T).contiguous()
        pce, nmc = torch.empty(B, T, device=th.device), torch.empty(B, T, device=th.device)
        ws = self._hist_ws.get(_lib.lib.aline_eig_history_workspace_bytes(L1, B, T), th.device)
        _lib.check(_lib.lib.aline_eig_ces_history(th.data_ptr(), xx.data_ptr(), yy.data_ptr(), L1, B, T, self.noise_scale, self.epsilon,
                                                  pce.data_ptr(), nmc.data_ptr(), None, ws.data_ptr(), ws.numel(),
                                                  _lib.stream_ptr(th.device)), "eig_ces_history")
        return pce, nmc

    @torch.no_grad()
    def sample_batch(self, batch_size):
        theta = self.sample_theta(batch_size).reshape(batch_size, self.n_theta, 1)
        x = self.sample_data(batch_size, self.n_context_init + self.n_query_init)
        y = self.forward(x, theta.squeeze(-1).unsqueeze(-2))
        batch = AttrDict()
        batch.context_x = x[:, :self.n_context_init].contiguous()
        batch.context_y = y[:, :self.n_context_init].contiguous()
        batch.query_x = x[:, self.n_context_init:].contiguous()
        batch.query_y = y[:, self.n_context_init:].contiguous()
        batch.target_all = batch.target_theta = theta
        batch.n_theta = self.n_theta
        return batch

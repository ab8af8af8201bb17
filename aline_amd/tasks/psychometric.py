"""Mirror of the reference's `PsychometricTask` sampler (tasks/psychometric.py:8-222): Gumbel
psychometric function with guess / lapse rates, Bernoulli outcomes.  The per-point Python loop of
sample_batch (psychometric.py:211-212) is vectorised on the device."""
import torch

from ..utils.attrdict import AttrDict
from .base_task import Task


class PsychometricTask(Task):
    def __init__(self, name: str = "Psychometric", dim_x: int = 1, dim_y: int = 1, embedding_type="theta",
                 n_target_theta: int = 4, n_context_init: int = 5, n_query_init: int = 300, design_scale: int = 5,
                 device=None, **kwargs) -> None:
        super().__init__(dim_x=dim_x, dim_y=dim_y, device=device)
        self.name, self.embedding_type = name, embedding_type
        self.n_target_theta, self.n_context_init, self.n_query_init = n_target_theta, n_context_init, n_query_init
        self.design_scale = design_scale
        self.bounds = ((-3.0, 3.0), (0.1, 2.0), (0.1, 0.9), (0.0, 0.5))       # alpha, beta, gamma, lambda

    @torch.no_grad()
    def sample_theta(self, batch_size):
        cols = [lo + (hi - lo) * torch.rand(batch_size, device=self.device) for lo, hi in self.bounds]
        return torch.stack(cols, dim=1).reshape(batch_size, 4, 1)

    @torch.no_grad()
    def sample_data(self, batch_size, n_data):
        return torch.rand(batch_size, n_data, self.dim_x, device=self.device) * 2 * self.design_scale - self.design_scale

    def unnormalise_design(self, x):
        return x

    def psychometric_function(self, x, theta):
        """x [B, N, 1], theta [B, 4, 1] -> p [B, N, 1] (psychometric.py:107-134)."""
        alpha, beta, gamma, lmbda = (theta[:, i:i + 1, :] for i in range(4))
        z = (x - alpha) / beta
        Fz = 1 - torch.exp(-10 ** z)
        return lmbda * gamma + (1 - lmbda) * Fz

    def forward(self, xi, theta):
        return torch.bernoulli(self.psychometric_function(xi, theta))

    def log_likelihood(self, y, xi, theta):
        p = self.psychometric_function(xi, theta)
        return y * torch.log(p + 1e-10) + (1 - y) * torch.log(1 - p + 1e-10)

    @torch.no_grad()
    def sample_batch(self, batch_size):
        theta = self.sample_theta(batch_size)
        n_c = self.n_context_init
        x = self.sample_data(batch_size, n_c + self.n_query_init)
        y = self.forward(x, theta)
        batch = AttrDict()
        batch.context_x, batch.context_y = x[:, :n_c].contiguous(), y[:, :n_c].contiguous()
        batch.query_x, batch.query_y = x[:, n_c:].contiguous(), y[:, n_c:].contiguous()
        batch.target_all = batch.target_theta = theta
        batch.n_target_theta = self.n_target_theta
        return batch

"""Mirror of the reference's `GPTask` sampler (tasks/gaussian_process.py:8-530), SURVEY.md 8-f.1:
the per-episode Python loop with one Cholesky per episode (gaussian_process.py:391-415) is replaced
by a batched kernel matrix + one batched Cholesky kernel on the device (C ABI `aline_cholesky_upper`;
input generator, not on the scored path)."""
import math

import torch

from .. import _lib
from ..utils.attrdict import AttrDict
from .base_task import Task


class GPTask(Task):
    KERNELS = ("rbf", "matern12", "matern32", "matern52")

    def __init__(self, name: str = "AL_mix", dim_x: int = 1, dim_y: int = 1, embedding_type="mix",
                 n_context_init: int = 5, n_query_init: int = 10, n_target_theta: int = 2, n_target_data: int = 5,
                 design_scale=None, noise_scale: float = 0.01, p_iso: float = 0.5, kernel_weights=None,
                 lengthscale_lower: float = 0.1, lengthscale_upper: float = 2.0, device=None, **kwargs) -> None:
        super().__init__(dim_x=dim_x, dim_y=dim_y, device=device)
        self.name, self.embedding_type = name, embedding_type
        self.n_context_init, self.n_query_init = n_context_init, n_query_init
        self.n_target_theta, self.n_target_data = n_target_theta, n_target_data
        self.jitter, self.p_iso = 1e-5, p_iso
        self.kernel_weights = kernel_weights if kernel_weights is not None else [1 / 3, 0, 1 / 3, 1 / 3]
        if embedding_type in ("mix", "theta"):
            if n_target_theta != dim_x + 1:                          # gaussian_process.py:64-66
                raise ValueError("n_target_theta must be equal to dim_x + 1 for theta or mix embedding type")
        else:
            self.n_target_theta = 0
        base = math.sqrt(dim_x)
        self.lengthscale_lower, self.lengthscale_upper = lengthscale_lower * base, lengthscale_upper * base
        self.scale_lower, self.scale_upper = 0.1, 1.0
        self.noise_scale = noise_scale
        self.design_scale = 5.0 if design_scale is None else float(design_scale)

    @torch.no_grad()
    def sample_theta(self, batch_size):
        dev = self.device
        ls = self.lengthscale_lower + (self.lengthscale_upper - self.lengthscale_lower) * torch.rand(
            batch_size, self.dim_x, device=dev)
        iso = torch.rand(batch_size, device=dev) < self.p_iso
        ls = torch.where(iso[:, None], ls[:, :1].expand_as(ls), ls)
        scale = self.scale_lower + (self.scale_upper - self.scale_lower) * torch.rand(batch_size, device=dev)
        return torch.cat([ls, scale[:, None]], dim=1).unsqueeze(2)             # [B, D+1, 1]

    @torch.no_grad()
    def sample_data(self, batch_size, n_data):
        return torch.rand(batch_size, n_data, self.dim_x, device=self.device) * 2 * self.design_scale - self.design_scale

    @torch.no_grad()
    def kernel_matrix(self, x, lengthscales, scale, ktype):
        """Batched K [B, N, N] for per-episode kernel types (0 rbf, 1 matern12, 2 matern32, 3 matern52)."""
        z = x / lengthscales[:, None, :]
        sq = (z.unsqueeze(2) - z.unsqueeze(1)).pow(2).sum(-1)                   # [B, N, N]
        r = torch.sqrt(sq)
        s3, s5 = math.sqrt(3.0), math.sqrt(5.0)
        k = torch.stack([torch.exp(-0.5 * sq), torch.exp(-r), (1 + s3 * r) * torch.exp(-s3 * r),
                         (1 + s5 * r + (5.0 / 3.0) * sq) * torch.exp(-s5 * r)], dim=0)    # [4, B, N, N]
        pick = k[ktype, torch.arange(x.shape[0], device=x.device)]
        return scale[:, None, None] * pick

    @torch.no_grad()
    def generate_gp_data(self, x, theta):
        B, n, _ = x.shape
        ls, scale = theta[:, :self.dim_x, 0], theta[:, self.dim_x, 0]
        w = torch.tensor(self.kernel_weights, dtype=torch.float, device=x.device)
        ktype = torch.multinomial(w / w.sum(), B, replacement=True)
        K = self.kernel_matrix(x, ls, scale, ktype) + self.jitter * torch.eye(n, device=x.device)
        K = K.contiguous()
        info = torch.zeros(1, dtype=torch.int32, device=x.device)
        _lib.check(_lib.lib.aline_cholesky_upper(K.data_ptr(), n, B, info.data_ptr(), _lib.stream_ptr(x.device)),
                   "cholesky_upper")                                           # K <- U, K = U^T U
        if int(info.item()) != 0:      # a non-positive pivot (fp32 RBF + jitter can lose definiteness): the reference's
            # torch.linalg.cholesky raises there (gaussian_process.py:407); a silently wrong factor is never returned
            raise RuntimeError("aline_amd: GP kernel matrix is not positive definite in fp32 (cholesky info != 0)")
        f = K.transpose(1, 2) @ torch.randn(B, n, 1, device=x.device)          # f = L z
        return f + self.noise_scale * torch.randn(B, n, 1, device=x.device)

    @torch.no_grad()
    def sample_batch(self, batch_size):
        batch = AttrDict()
        theta = self.sample_theta(batch_size)
        n_c, n_q = self.n_context_init, self.n_query_init
        n_td = 0 if self.embedding_type == "theta" else self.n_target_data
        x = self.sample_data(batch_size, n_c + n_q + n_td)
        y = self.generate_gp_data(x, theta)
        batch.context_x, batch.context_y = x[:, :n_c].contiguous(), y[:, :n_c].contiguous()
        batch.query_x, batch.query_y = x[:, n_c:n_c + n_q].contiguous(), y[:, n_c:n_c + n_q].contiguous()
        if self.embedding_type == "theta":
            batch.target_all = batch.target_theta = theta
            batch.target_x = batch.target_y = None
        else:
            batch.target_x, batch.target_y = x[:, n_c + n_q:].contiguous(), y[:, n_c + n_q:].contiguous()
            if self.embedding_type == "data":
                batch.target_all, batch.target_theta = batch.target_y, None
            else:
                batch.target_theta = theta
                batch.target_all = torch.cat([batch.target_y, theta], dim=1)   # data first, theta last (:526)
        batch.n_target_theta = self.n_target_theta
        return batch

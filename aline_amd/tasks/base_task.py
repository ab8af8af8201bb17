"""Mirror of the reference's tasks/base_task.py (Task: design-space maps :58-72, update_batch
:103-154).  `update_batch` keeps the reference's shape-changing semantics for callers that step
the model from Python; the fast path never calls it (roles are updated in-kernel, see rollout.py)."""
import torch
import torch.nn as nn


class Task(nn.Module):
    def __init__(self, dim_x: int = 2, dim_y: int = 1, design_scale: float = 1.0,
                 outcome_scale: float = 1.0, device=None, **kwargs) -> None:
        super().__init__()
        self.dim_x, self.dim_y = dim_x, dim_y
        self.design_scale, self.outcome_scale = design_scale, outcome_scale
        self.device = torch.device(device if device is not None else "cuda")

    def to_design_space(self, xi):
        return xi * self.design_scale

    def normalise_design(self, x):
        return x / self.design_scale

    def unnormalise_design(self, x):
        return x * self.design_scale

    def normalise_outcomes(self, y):
        return y / self.outcome_scale

    def update_batch_query(self, query, idx):
        B, Nt, D = query.shape
        mask = torch.ones((B, Nt), dtype=torch.bool, device=query.device)
        mask[torch.arange(B, device=query.device).unsqueeze(1), idx] = False
        return query[mask].view(B, -1, D)

    def update_batch_context(self, context, new):
        return torch.cat([context, new], dim=1)

    def update_batch(self, batch, idx):
        next_x = torch.gather(batch.query_x, 1, idx.unsqueeze(2).expand(-1, 1, self.dim_x))
        next_y = torch.gather(batch.query_y, 1, idx.unsqueeze(2).expand(-1, 1, self.dim_y))
        batch.query_x = self.update_batch_query(batch.query_x, idx)
        batch.query_y = self.update_batch_query(batch.query_y, idx)
        batch.context_x = self.update_batch_context(batch.context_x, next_x)
        batch.context_y = self.update_batch_context(batch.context_y, next_y)
        return batch

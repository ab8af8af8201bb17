"""Mirror of the reference's `EIGStepLoss` (loss/eig.py:154-209): sequential accumulation of
log p(y_t | xi_t, theta_l) over L+1 contrastive samples and the sPCE / sNMC logsumexps, on the HIP
kernels.  `log_prob` must be a task with a native likelihood (HiddenLocation, CESTask)."""
import torch
import torch.nn as nn

from .. import _lib


class EIGStepLoss(nn.Module):
    def __init__(self, L: int, M: int, log_prob, reduction=None, device=None) -> None:
        super().__init__()
        self.L, self.M, self.reduction = L, M, reduction
        self.task = getattr(log_prob, "__self__", log_prob)   # accept task or task.log_likelihood
        if not hasattr(self.task, "native_eig_step"):
            raise NotImplementedError("aline_amd: EIGStepLoss needs a task with a native likelihood")
        self.device = torch.device(device if device is not None else "cuda")
        self.seq_logprobs = torch.zeros((L + 1, M), device=self.device)
        self._ws = _lib.Workspace()

    def reset(self):
        self.seq_logprobs.zero_()

    def step(self, y_outcomes, xi_designs, thetas):
        """y [M, D_y], xi [M, D_x], thetas [L+1, M, ...] -> running S [L+1, M] (eig.py:174-193)."""
        self.task.native_eig_step(_lib.f32(thetas), _lib.f32(xi_designs), _lib.f32(y_outcomes),
                                  self.seq_logprobs)
        return self.seq_logprobs

    def forward(self, y_outcomes, xi_designs, thetas):
        S = self.step(y_outcomes, xi_designs, thetas)
        L1, B = S.shape
        pce_b, nmc_b = torch.empty(B, device=S.device), torch.empty(B, device=S.device)
        nb = _lib.lib.aline_eig_finalize_workspace_bytes(L1, B)
        ws = self._ws.get(nb, S.device)
        _lib.check(_lib.lib.aline_eig_finalize(S.data_ptr(), L1, B, pce_b.data_ptr(), nmc_b.data_ptr(),
                                               ws.data_ptr(), ws.numel(), _lib.stream_ptr(S.device)),
                   "eig_finalize")
        # the kernel returns the bounds log(L+1) - pce_loss / log L - nmc_loss; the reference class
        # returns the losses (eig.py:200-202), so undo the constants here
        import math
        pce_loss = math.log(L1) - pce_b
        nmc_loss = math.log(L1 - 1) - nmc_b
        if self.reduction == "mean":
            pce_loss, nmc_loss = pce_loss.mean(), nmc_loss.mean()
        return pce_loss, nmc_loss

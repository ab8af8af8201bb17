"""GPU: device-side task samplers (input generators) produce batches of the
reference's shape contract, statistically consistent with the simulators, and the whole path runs on
them (cfg1/cfg3/cfg5 geometries through the generic pipeline)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_gp_task_batch_contract_and_statistics():
    from aline_amd.tasks import GPTask
    torch.manual_seed(0)
    task = GPTask(dim_x=2, embedding_type="mix", n_context_init=1, n_query_init=200, n_target_theta=3,
                  n_target_data=100, design_scale=5, noise_scale=0.01)
    b = task.sample_batch(64)
    assert b.context_x.shape == (64, 1, 2) and b.query_y.shape == (64, 200, 1)
    assert b.target_x.shape == (64, 100, 2) and b.target_all.shape == (64, 103, 1)
    assert torch.equal(b.target_all[:, :100], b.target_y) and torch.equal(b.target_all[:, 100:], b.target_theta)
    assert float(b.query_x.abs().max()) <= 5.0
    # marginal variance of a GP draw = output scale (+ noise): check the batch average
    y = torch.cat([b.context_y, b.query_y, b.target_y], 1)
    ratio = (y.var(dim=1).squeeze(-1) / b.target_theta[:, 2, 0]).mean()
    assert 0.5 < float(ratio) < 1.5
    th = b.target_theta[:, :, 0]
    assert float(th[:, :2].min()) >= 0.1 * math.sqrt(2) - 1e-6 and float(th[:, 2].max()) <= 1.0


@pytest.mark.parametrize("B,n", [(7, 301), (3, 1), (5, 8), (4, 9), (3, 256), (2, 257), (2, 520), (1, 1100)])
def test_batched_cholesky_kernel(B, n):
    """Blocked kernel (8 rows per sweep; 1 / 2 / 4 columns per thread) and the row-wise fallback (n > 1024)."""
    from aline_amd import _lib
    torch.manual_seed(1)
    X = torch.randn(B, n, 40, device="cuda")
    A = X @ X.transpose(1, 2) / 40 + 0.5 * torch.eye(n, device="cuda")
    U = A.clone().contiguous()
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    _lib.check(_lib.lib.aline_cholesky_upper(U.data_ptr(), n, B, info.data_ptr(), _lib.stream_ptr(U.device)), "chol")
    torch.cuda.synchronize()
    assert int(info) == 0
    assert float(torch.tril(U, -1).abs().max()) == 0.0
    rec = U.transpose(1, 2) @ U
    assert float((rec - A).abs().max()) < 2e-4 * max(1.0, n / 301)
    ref = torch.linalg.cholesky(A.double().cpu())               # CPU LAPACK as the checker
    assert float((U.transpose(1, 2).cpu().double() - ref).abs().max()) < 1e-3


def test_batched_cholesky_flags_a_non_positive_matrix():
    from aline_amd import _lib
    A = torch.eye(20, device="cuda").repeat(3, 1, 1).contiguous()
    A[1, 11, 11] = -1.0
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    _lib.check(_lib.lib.aline_cholesky_upper(A.data_ptr(), 20, 3, info.data_ptr(), _lib.stream_ptr(A.device)), "chol")
    torch.cuda.synchronize()
    assert int(info) == 1 and torch.isfinite(A).all()


def test_psychometric_task_and_model_run():
    from aline_amd import Aline, Embedder, Encoder, OutputHead
    from aline_amd.rollout import Rollout
    from aline_amd.tasks import PsychometricTask
    from aline_amd.utils import create_target_mask
    torch.manual_seed(0)
    task = PsychometricTask(n_context_init=1, n_query_init=200)
    b = task.sample_batch(32)
    assert set(torch.unique(b.query_y).tolist()) <= {0.0, 1.0}
    assert b.target_all.shape == (32, 4, 1)
    p = task.psychometric_function(b.query_x, b.target_theta)
    assert float(p.min()) >= 0.0 and float(p.max()) <= 1.0
    b.target_mask = create_target_mask("predefined", "theta", 0, 4, None,
                                       [[False, False, True, True], [True, True, False, False]], None, 0, None)
    model = Aline(Embedder(1, 1, 64, 128, 4, "theta"), Encoder(64, 128, 8, 0.0, 2), OutputHead(1, 1, 64, 128)).cuda()
    ro = Rollout(model, b, 5, select="sample").run()
    torch.cuda.synchronize()
    assert torch.isfinite(ro.target_ll).all() and ro.target_ll.shape == (5, 32, 4)


def test_eval_boed_small():
    from aline_amd import Aline, Embedder, Encoder, OutputHead
    from aline_amd.tasks import HiddenLocation
    from aline_amd.utils import eval_boed
    torch.manual_seed(0)
    model = Aline(Embedder(2, 1, 32, 128, 2, "theta"), Encoder(32, 128, 4, 0.0, 3), OutputHead(2, 1, 32, 128)).cuda()
    out = eval_boed(model, HiddenLocation(n_query_init=30), T=4, L=5000, M=24, batch_size=8, stepwise=True)
    assert out.pce_mean.shape == (5,) and torch.isfinite(out.pce_mean).all() and torch.isfinite(out.nmc_err).all()

"""GPU: edge cases of the acquisition loop through the C ABI against the CPU oracle -- the query set exhausted down to
one candidate, a target mask with nothing selected (queries see the context only, encoder.py:110-124), a single
episode, several initial context points; fused (d=32) and generic pipelines."""
import os

import pytest
import torch

import aline_oracle as orc
from helpers import maxdiff, native_model, to_dev

pytestmark = pytest.mark.gpu

DIMS = dict(dim_x=2, dim_y=1, d=32, F=128, n_head=4, L=3, C=10, n_theta=2, embedding_type="theta", time_token=False)
CFG = dict(embedding_type="theta", n_head=4, num_layers=3, num_components=10, std_min=1e-4, n_target_theta=2)


def _batch(B, n_c, n_q, seed, mask=None):
    g = torch.Generator().manual_seed(seed)
    b = dict(context_x=torch.rand(B, n_c, 2, generator=g), context_y=torch.randn(B, n_c, 1, generator=g),
             query_x=torch.rand(B, n_q, 2, generator=g), query_y=torch.randn(B, n_q, 1, generator=g),
             target_all=torch.rand(B, 2, 1, generator=g))
    b["target_theta"] = b["target_all"]
    if mask is not None:
        b["target_mask"] = torch.tensor(mask)
    return b, g


@pytest.mark.parametrize("B,n_c,n_q,T,mask", [
    (1, 1, 9, 9, None),              # one episode, the query set runs out: the last step has a single candidate
    (3, 4, 17, 17, [False, False]),  # nothing selected: query rows see the context only
    (5, 2, 33, 12, [True, False]),   # ragged tile (2 + 33 + 2 = 37 rows), one selected target
])
@pytest.mark.parametrize("fused", [True, False])
def test_exhausted_queries_masks_and_small_batches(B, n_c, n_q, T, mask, fused):
    from aline_amd.rollout import Rollout
    model, sd = native_model(DIMS, 7)
    batch, g = _batch(B, n_c, n_q, 11 * B + n_q, mask)
    forced = torch.stack([torch.randint(0, n_q - t, (B,), generator=g) for t in range(T)], 1)
    ref = orc.rollout(sd, batch, CFG, T, forced_idx=forced, mask_type="all" if mask is None else "partial")
    from aline_amd import _lib
    with _lib.debug(*([] if fused else ["DISABLE_FUSED"])):
        ro = Rollout(model, to_dev(batch), T, select="forced", forced_idx=forced, keep_zt=True).run()
        torch.cuda.synchronize()
    assert maxdiff(ro.target_ll, torch.stack(ref["target_ll"])) < 1e-4
    assert maxdiff(ro.log_prob, torch.stack(ref["log_prob"], 1)) < 1e-4
    zt = ro.zt.cpu()                                     # [T, B, n_q] zero padded
    for t in range(T):
        assert torch.allclose(zt[t, :, :n_q - t].sum(-1), torch.ones(B), atol=1e-5)
        if t > 0:
            assert float(zt[t, :, n_q - t:].abs().max()) == 0.0
    if T == n_q:                                          # single remaining candidate: probability one, log_prob zero
        assert torch.allclose(zt[T - 1, :, 0], torch.ones(B), atol=1e-6)
        assert float(ro.log_prob[:, T - 1].abs().max()) < 1e-6
    # the exported context equals the order of entry (update_batch, base_task.py:133-154)
    cx, cy = ro.export_context()
    assert torch.equal(cx.cpu(), ref["batch"]["context_x"]) and torch.equal(cy.cpu(), ref["batch"]["context_y"])


def test_argmax_and_sample_modes_on_one_candidate():
    """n_query = 1: every selection mode must pick index 0 with log-probability 0."""
    from aline_amd.rollout import Rollout
    model, _ = native_model(DIMS, 3)
    batch, _ = _batch(4, 2, 1, 5)
    for mode in ("argmax", "sample"):
        model.train(mode == "sample")
        ro = Rollout(model, to_dev(batch), 1, select=mode).run()
        torch.cuda.synchronize()
        assert int(ro.idx.abs().max()) == 0 and float(ro.log_prob.abs().max()) < 1e-6
        assert torch.isfinite(ro.target_ll).all()

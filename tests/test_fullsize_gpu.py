"""GPU, BASELINE.json full size (location_finding, B=1000, T=30, n_query=200): size-independent
properties of the rollout, and agreement of the fused kernel with the generic per-op pipeline."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _model_and_batch(B=1000, n_query=200, seed=0):
    from aline_amd import Aline, Embedder, Encoder, OutputHead
    from aline_amd.tasks import HiddenLocation
    torch.manual_seed(seed)
    model = Aline(Embedder(2, 1, 32, 128, 2, "theta"), Encoder(32, 128, 4, 0.0, 3),
                  OutputHead(2, 1, 32, 128)).cuda()
    # non-trivial LayerNorm / bias values so that every parameter matters
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.05 * torch.randn_like(p))
    task = HiddenLocation(n_query_init=n_query)
    return model, task.sample_batch(B)


def _rollout(model, batch, T, fused, **kw):
    from aline_amd import _lib
    from aline_amd.rollout import Rollout
    with _lib.debug(*([] if fused else ["DISABLE_FUSED"])):
        ro = Rollout(model, batch, T, **kw).run()
        torch.cuda.synchronize()
    return ro


def test_fullsize_rollout_invariants_and_fused_vs_generic():
    B, T, nq = 1000, 30, 200
    model, batch = _model_and_batch(B, nq)
    model.eval()
    fu = _rollout(model, batch, T, True, select="argmax", keep_zt=True)
    # structural invariants of the acquisition loop (train_aline.py:80-110, base_task.py:133-154)
    role = fu.role.cpu()
    assert (role > 0).sum(1).eq(1 + T).all()                          # one new context point per step
    for b in (0, 17, 999):
        assert sorted(role[b][role[b] > 0].tolist()) == list(range(1, T + 2))   # orders 1..T+1, once each
    slot = fu.slot.cpu()
    assert all(len(set(slot[b].tolist())) == T for b in range(0, B, 97))          # a design is chosen once
    idx = fu.idx.cpu()
    assert (idx >= 0).all() and (idx < torch.arange(nq, nq - T, -1)[None]).all()   # index into the shrinking list
    zt = fu.zt.cpu()                                                   # [T, B, nq] zero padded
    assert torch.allclose(zt.sum(-1), torch.ones(T, B), atol=1e-5)
    for t in (1, 13, T - 1):
        assert float(zt[t, :, nq - t:].abs().max()) == 0.0
    assert torch.isfinite(fu.target_ll).all() and torch.isfinite(fu.log_prob).all()
    # fused kernel == generic pipeline (teacher-forced with the fused designs), all 1000 episodes
    ge = _rollout(model, batch, T, False, select="forced", forced_idx=fu.idx, keep_zt=True)
    assert float((ge.target_ll - fu.target_ll).abs().max()) < 1e-4    # NLL bound of the north star
    assert float((ge.log_prob - fu.log_prob).abs().max()) < 2e-4
    assert float((ge.zt - fu.zt).abs().max()) < 5e-5
    assert (ge.slot == fu.slot).all()
    # free-running generic argmax picks the same designs almost everywhere (fp-order ties excepted)
    gf = _rollout(model, batch, T, False, select="argmax")
    assert float((gf.idx == fu.idx).float().mean()) > 0.98


def test_query_permutation_equivariance():
    """Set attention has no positional encoding (encoder.py:83-126): permuting the candidate designs
    permutes their scores and leaves the posterior unchanged."""
    B, nq = 64, 200
    model, batch = _model_and_batch(B, nq, seed=1)
    model.eval()
    from aline_amd.utils import AttrDict
    perm = torch.randperm(nq, device="cuda")
    pb = AttrDict(dict(batch))
    pb.query_x, pb.query_y = batch.query_x[:, perm].contiguous(), batch.query_y[:, perm].contiguous()
    a = _rollout(model, batch, 1, True, select="argmax", keep_zt=True)
    b = _rollout(model, pb, 1, True, select="argmax", keep_zt=True)
    assert float((a.zt[0][:, perm] - b.zt[0]).abs().max()) < 2e-6
    assert float((a.target_ll - b.target_ll).abs().max()) < 2e-5
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(nq, device="cuda")
    assert (inv[a.idx[:, 0]] == b.idx[:, 0]).float().mean() > 0.97


def test_partial_workgroup_and_small_batches():
    """B not a multiple of the 4 episodes per workgroup, and B < 4."""
    for B in (1, 3, 5, 7):
        model, batch = _model_and_batch(B, 40, seed=B)
        model.eval()
        fu = _rollout(model, batch, 5, True, select="argmax")
        ge = _rollout(model, batch, 5, False, select="forced", forced_idx=fu.idx)
        assert float((ge.target_ll - fu.target_ll).abs().max()) < 1e-4


def test_fullsize_s3_headline_matches_exact_fp32_kernel():
    """The benchmarked path (precision f16x3 -> s3::step_kernel, 3-term f16 split) against round 1's exact-fp32 fused
    kernel on all 1000 episodes x 30 steps of the headline config, teacher-forced with the designs s3 picked."""
    B, T, nq = 1000, 30, 200
    model, batch = _model_and_batch(B, nq)
    model.eval()
    model.set_precision("f16x3")
    s3 = _rollout(model, batch, T, True, select="argmax", keep_zt=True)
    assert s3.path == "s3::step_kernel"
    role = s3.role.cpu()
    assert (role > 0).sum(1).eq(1 + T).all()
    zt = s3.zt.cpu()
    assert torch.allclose(zt.sum(-1), torch.ones(T, B), atol=1e-5)
    model.set_precision("f32")
    fu = _rollout(model, batch, T, True, select="forced", forced_idx=s3.idx, keep_zt=True)
    assert fu.path == "fused::rollout_f32_kernel"
    assert float((fu.target_ll - s3.target_ll).abs().max()) < 1e-4    # NLL bound of the north star, every target
    assert float((fu.log_prob - s3.log_prob).abs().max()) < 2e-4
    assert float((fu.zt - s3.zt).abs().max()) < 5e-5
    assert (fu.slot == s3.slot).all()
    free = _rollout(model, batch, T, True, select="argmax")
    assert float((free.idx == s3.idx).float().mean()) > 0.98          # (fp-order ties excepted)


def test_fullsize_cfg3_s3_matches_generic_fp32():
    """BASELINE configs[2] per GPU (al_mix dx = 2, B = 512 of 4096, T = 50, n_query = 200, 100 data + 3 theta targets,
    split mask on the data targets: up to 150 keys, 304 token rows): s3 against the generic exact-fp32 pipeline."""
    from aline_amd import Aline, Embedder, Encoder, OutputHead
    from aline_amd.rollout import Rollout
    from aline_amd.tasks import GPTask
    from aline_amd.utils import create_target_mask
    B, T, nq = 512, 50, 200
    torch.manual_seed(2)
    model = Aline(Embedder(2, 1, 32, 128, 3, "mix"), Encoder(32, 128, 4, 0.0, 3), OutputHead(2, 1, 32, 128)).cuda().eval()
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.05 * torch.randn_like(p))
    task = GPTask(dim_x=2, embedding_type="mix", n_context_init=1, n_query_init=nq, n_target_theta=3, n_target_data=100,
                  device=torch.device("cuda"))
    batch = task.sample_batch(B)
    batch["target_mask"] = create_target_mask("split", "mix", 100, 3, None, None, None, None, "data")
    model.set_precision("f16x3")
    s3 = Rollout(model, batch, T, select="argmax", keep_zt=True)
    assert s3.path == "s3::step_kernel"
    s3.run()
    model.set_precision("f32")
    ge = Rollout(model, batch, T, select="forced", forced_idx=s3.idx, keep_zt=True)
    assert ge.path == "generic pipeline"
    ge.run()
    torch.cuda.synchronize()
    assert torch.isfinite(s3.target_ll).all()
    assert (s3.role.cpu() > 0).sum(1).eq(1 + T).all()
    d = (ge.target_ll - s3.target_ll).abs()
    assert float(d.mean(-1).max()) < 1e-4 and float(d.max()) < 5e-4, (float(d.mean(-1).max()), float(d.max()))
    assert float((ge.log_prob - s3.log_prob).abs().max()) < 2e-4
    assert (ge.slot == s3.slot).all()


def _oracle_slice(model, batch, ro, T, cfg, take, mask_type="all"):
    """The CPU oracle on the episodes `take` of a full-size batch, teacher-forced with the designs the HIP rollout chose for
    them (episodes are independent: the slice is an exact check of those rows of the full-size run)."""
    import aline_oracle as orc
    sd = orc.cast_state_dict(model.state_dict())
    cpu = {k: v[take].cpu() for k, v in batch.items() if torch.is_tensor(v) and v.dim() >= 2 and v.shape[0] == ro.B}
    if batch.get("target_mask") is not None:
        cpu["target_mask"] = batch["target_mask"].cpu()
    return orc.rollout(sd, cpu, cfg, T, forced_idx=ro.idx[take].cpu(), mask_type=mask_type)


@pytest.mark.parametrize("precision,path", [("f16x3", "s3::step_kernel"), ("f32", "fused::rollout_f32_kernel")])
def test_fullsize_headline_slice_against_the_cpu_oracle(precision, path):
    """BASELINE configs[1] at full size (B = 1000, T = 30, n_query = 200): 16 of the 1000 episodes (first, last, spread)
    through the oracle, teacher-forced with the designs of the full-size HIP rollout."""
    B, T, nq = 1000, 30, 200
    model, batch = _model_and_batch(B, nq)
    model.eval()
    model.set_precision(precision)
    ro = _rollout(model, batch, T, True, select="argmax", keep_zt=True)
    assert ro.path == path and ro.range_status() == 0
    take = torch.tensor([0, 1, 2, 3, 63, 64, 250, 251, 499, 500, 777, 778, 996, 997, 998, 999])
    cfg = dict(embedding_type="theta", n_head=4, num_layers=3, num_components=10, std_min=1e-4, n_target_theta=2)
    ref = _oracle_slice(model, batch, ro, T, cfg, take)
    assert float((ro.target_ll[:, take].cpu() - torch.stack(ref["target_ll"])).abs().max()) < 1e-4
    assert float((ro.log_prob[take].cpu() - torch.stack(ref["log_prob"], 1)).abs().max()) < 2e-4
    for t in (0, 7, T - 1):
        assert float((ro.zt[t][take].cpu()[:, :nq - t] - ref["zt"][t]).abs().max()) < 5e-5


def test_fullsize_cfg3_slice_against_the_cpu_oracle():
    """BASELINE configs[2] per GPU (al_mix dx = 2, B = 512, T = 50, n_query = 200, split mask: up to 150 keys, 304 rows):
    16 of the 512 episodes of the full-size s3 rollout through the oracle."""
    from aline_amd import Aline, Embedder, Encoder, OutputHead
    from aline_amd.rollout import Rollout
    from aline_amd.tasks import GPTask
    from aline_amd.utils import create_target_mask
    B, T, nq = 512, 50, 200
    torch.manual_seed(2)
    model = Aline(Embedder(2, 1, 32, 128, 3, "mix"), Encoder(32, 128, 4, 0.0, 3), OutputHead(2, 1, 32, 128)).cuda().eval()
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.05 * torch.randn_like(p))
    task = GPTask(dim_x=2, embedding_type="mix", n_context_init=1, n_query_init=nq, n_target_theta=3, n_target_data=100,
                  device=torch.device("cuda"))
    batch = task.sample_batch(B)
    batch["target_mask"] = create_target_mask("split", "mix", 100, 3, None, None, None, None, "data")
    model.set_precision("f16x3")
    ro = Rollout(model, batch, T, select="argmax", keep_zt=True)
    assert ro.path == "s3::step_kernel"
    ro.run()
    torch.cuda.synchronize()
    assert ro.range_status() == 0
    take = torch.tensor([0, 1, 2, 3, 100, 101, 255, 256, 257, 300, 400, 401, 508, 509, 510, 511])
    cfg = dict(embedding_type="mix", n_head=4, num_layers=3, num_components=10, std_min=1e-4, n_target_theta=3)
    ref = _oracle_slice(model, batch, ro, T, cfg, take, mask_type="split")
    d = (ro.target_ll[:, take].cpu() - torch.stack(ref["target_ll"])).abs()
    # NLL (the mean over an episode's targets, what train_aline.py:97-110 reduces) within 1e-4; single targets of the untrained
    # GP model with sharp mixture components within 5e-4 (test_s3_gpu.py explains the amplification)
    assert float(d.mean(-1).max()) < 1e-4 and float(d.max()) < 5e-4, (float(d.mean(-1).max()), float(d.max()))
    assert float((ro.log_prob[take].cpu() - torch.stack(ref["log_prob"], 1)).abs().max()) < 2e-4


@pytest.mark.parametrize("mask", [[False, False, True, True], [True, True, False, False]])
def test_fullsize_cfg5_slice_against_the_cpu_oracle(mask):
    """BASELINE configs[4] (psychometric, d_model = 512 / 8 heads of 64, F = 128 as config/encoder/encoder.yaml, predefined target
    masks of config/task/psychometric.yaml:12, T = 30, n_query = 200: 205 rows, up to 32 keys) at B = 256 on the x5 path: 8 of the 256
    episodes through the CPU oracle, teacher-forced with the designs of the full-size HIP rollout."""
    from aline_amd import Aline, Embedder, Encoder, OutputHead
    from aline_amd.rollout import Rollout
    from aline_amd.tasks import PsychometricTask
    B, T, nq = 256, 30, 200
    torch.manual_seed(5)
    model = Aline(Embedder(1, 1, 512, 128, 4, "theta"), Encoder(512, 128, 8, 0.0, 3), OutputHead(1, 1, 512, 128)).cuda().eval()
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.02 * torch.randn_like(p))
    batch = PsychometricTask(n_query_init=nq, n_context_init=1, device=torch.device("cuda")).sample_batch(B)
    batch["target_mask"] = torch.tensor(mask)
    model.set_precision("f16x3")
    ro = Rollout(model, batch, T, select="argmax", keep_zt=True)
    assert ro.path == "x5::layer_kernel"
    ro.run()
    torch.cuda.synchronize()
    assert ro.range_status() == 0
    take = torch.tensor([0, 1, 2, 127, 128, 129, 254, 255])
    cfg = dict(embedding_type="theta", n_head=8, num_layers=3, num_components=10, std_min=1e-4, n_target_theta=4)
    ref = _oracle_slice(model, batch, ro, T, cfg, take)
    sel = torch.tensor(mask)
    d = (ro.target_ll[:, take].cpu() - torch.stack(ref["target_ll"])).abs()
    assert float(d[..., sel].max()) < 1e-4, float(d[..., sel].max())       # the selected targets: what the loss reads (train_aline.py:97-110)
    assert float(d.max()) < 5e-4, float(d.max())
    assert float((ro.log_prob[take].cpu() - torch.stack(ref["log_prob"], 1)).abs().max()) < 2e-4
    for t in (0, 15, T - 1):
        assert float((ro.zt[t][take].cpu()[:, :nq - t] - ref["zt"][t]).abs().max()) < 5e-5


def test_evaluation_protocol_size_against_the_cpu_oracle():
    """The README's evaluation runs n_query_final = 2000 candidates for T_final = 35 steps (README.md:45): P = 2001 point slots,
    126 token tiles per episode.  Eval-mode rollout on B = 2: the benchmarked mode (f16x3, s3 path) against the CPU oracle,
    teacher-forced with its own designs; the exact-fp32 mode against the same designs."""
    import aline_oracle as orc
    from aline_amd.rollout import Rollout
    B, T, nq = 2, 35, 2000
    model, batch = _model_and_batch(B, nq, seed=4)
    model.eval()
    model.set_precision("f16x3")
    free = Rollout(model, batch, T, select="argmax", keep_zt=True).run()
    torch.cuda.synchronize()
    assert free.path == "s3::step_kernel" and free.range_status() == 0
    sd = orc.cast_state_dict(model.state_dict())
    cfg = dict(embedding_type="theta", n_head=4, num_layers=3, num_components=10, std_min=1e-4, n_target_theta=2)
    cpu = {k: v.cpu() for k, v in batch.items() if torch.is_tensor(v)}
    ref = orc.rollout(sd, cpu, cfg, T, forced_idx=free.idx.cpu())
    rll, rlp = torch.stack(ref["target_ll"]), torch.stack(ref["log_prob"], 1)
    assert float((free.target_ll.cpu() - rll).abs().max()) < 1e-4
    assert float((free.log_prob.cpu() - rlp).abs().max()) < 2e-4
    assert float((free.zt[T - 1].cpu()[:, :nq - T + 1] - ref["zt"][T - 1]).abs().max()) < 5e-5
    zt = free.zt.cpu()
    assert torch.allclose(zt.sum(-1), torch.ones(T, B), atol=1e-5)
    cx, cy = free.export_context()
    assert cx.shape == (B, 1 + T, 2)
    # the oracle's argmax agrees with the designs the kernel chose (probability of the chosen design within 1e-6 of the row's maximum:
    # fp-order ties excepted, nothing else)
    for t in (0, T // 2, T - 1):
        z = ref["zt"][t]
        chosen = z.gather(1, free.idx[:, t:t + 1].cpu())
        assert float((z.max(-1, keepdim=True).values - chosen).max()) < 1e-6
    model.set_precision("f32")
    f32 = Rollout(model, batch, T, select="forced", forced_idx=free.idx, keep_zt=True).run()
    torch.cuda.synchronize()
    assert float((f32.target_ll.cpu() - rll).abs().max()) < 1e-4
    assert float((f32.log_prob.cpu() - rlp).abs().max()) < 2e-4

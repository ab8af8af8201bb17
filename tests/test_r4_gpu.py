"""GPU, round 4: the parity nets VERDICT r3 asked for -- the benchmarked d = 256 launch shape of `x3::layer_kernel` (full tile
rounds + a tail round of `ktail` < 8 waves + idle-streaming waves) against the CPU oracle, and `eval_boed` (utils/eval.py:142-198)
end to end on a tape of the reference's own random draws."""
import pytest
import torch

import aline_oracle as orc
from helpers import maxdiff, native_model

pytestmark = pytest.mark.gpu


def _d256_model(seed=0):
    from aline_amd import Aline, Embedder, Encoder, OutputHead
    torch.manual_seed(seed)
    model = Aline(Embedder(2, 1, 256, 1024, 2, "theta"), Encoder(256, 1024, 8, 0.0, 3), OutputHead(2, 1, 256, 1024)).cuda().eval()
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.02 * torch.randn_like(p))
    return model


def _x3_round_shape(B, tpe, n_cu=None):
    """The three-way round structure of x3::layer_kernel for B episodes of `tpe` tiles (x3_impl.h, tile rounds): (full, rem, ktail)."""
    n_cu = n_cu or torch.cuda.get_device_properties(0).multi_processor_count
    ntiles, per_round = B * tpe, n_cu * 8
    full = ntiles // per_round
    rem = ntiles - full * per_round
    return full, rem, (rem + n_cu - 1) // n_cu


@pytest.mark.parametrize("B,T,take", [
    # the bench's d256 leg (bench.py: B = 1000, T = 30, n_query = 200: 13 000 tiles = 6 full rounds of 2 048, 712 tail tiles on
    # 3 waves per workgroup, 5 idle-streaming waves): first / last / spread episodes, incl. episodes whose tiles all fall in the
    # tail round (tile >= 6 * 2048 = 12 288 <=> episode >= 946)
    (1000, 30, [0, 1, 157, 158, 314, 472, 473, 630, 787, 945, 946, 947, 970, 997, 998, 999]),
    # a mid-size case: 170 episodes = 2 210 tiles = 1 full round + 162 tail tiles (ktail = 1 on 256 CUs)
    (170, 6, [0, 1, 78, 79, 156, 157, 158, 159, 160, 165, 168, 169]),
])
def test_fullsize_d256_slice_against_the_cpu_oracle(B, T, take):
    from aline_amd.rollout import Rollout
    from aline_amd.tasks import HiddenLocation
    nq = 200
    model = _d256_model()
    torch.manual_seed(1)
    batch = HiddenLocation(n_query_init=nq).sample_batch(B)
    model.set_precision("f16x3")
    ro = Rollout(model, batch, T, select="argmax", keep_zt=True)
    assert ro.path == "x3::layer_kernel"
    ro.run()
    torch.cuda.synchronize()
    assert ro.range_status() == 0
    full, rem, ktail = _x3_round_shape(B, 13)
    assert full >= 1 and rem > 0 and ktail < 8, (full, rem, ktail)        # the branch combination this test is for
    take = torch.tensor(take)
    assert int(take.max()) * 13 + 12 >= full * torch.cuda.get_device_properties(0).multi_processor_count * 8   # a tail-round episode is in the slice
    sd = orc.cast_state_dict(model.state_dict())
    cpu = {k: v[take].cpu() for k, v in batch.items() if torch.is_tensor(v) and v.dim() >= 2 and v.shape[0] == B}
    cfg = dict(embedding_type="theta", n_head=8, num_layers=3, num_components=10, std_min=1e-4, n_target_theta=2)
    ref = orc.rollout(sd, cpu, cfg, T, forced_idx=ro.idx[take].cpu())
    assert float((ro.target_ll[:, take].cpu() - torch.stack(ref["target_ll"])).abs().max()) < 1e-4
    assert float((ro.log_prob[take].cpu() - torch.stack(ref["log_prob"], 1)).abs().max()) < 2e-4
    for t in (0, T // 2, T - 1):
        assert float((ro.zt[t][take].cpu()[:, :nq - t] - ref["zt"][t]).abs().max()) < 5e-5
    # every episode of the batch (not only the slice): finite, one new context point per step, probabilities sum to one
    assert torch.isfinite(ro.target_ll).all() and torch.isfinite(ro.log_prob).all()
    assert (ro.role.cpu() > 0).sum(1).eq(1 + T).all()
    assert torch.allclose(ro.zt.sum(-1).cpu(), torch.ones(T, B), atol=1e-5)


class _Tape:
    """Replays the recorded draws of the reference's `eval_boed` call through the product task's `sample_batch` / `sample_theta`."""

    def __init__(self, fx, task):
        from aline_amd.utils import AttrDict
        self.fx, self.nb, self.nt, self.AttrDict = fx, 0, 0, AttrDict
        task.sample_batch, task.sample_theta = self.sample_batch, self.sample_theta

    def sample_batch(self, batch_size):
        i, fx = self.nb, self.fx
        self.nb += 1
        self.nt += 1                  # (the reference's sample_batch draws its theta through sample_theta: that draw is in the batch)
        b = self.AttrDict({k: fx.t(f"batch{i}.{k}").cuda() for k in ("context_x", "context_y", "query_x", "query_y", "target_theta", "target_all")})
        assert b.context_x.shape[0] == batch_size
        b.n_target_theta = 2
        return b

    def sample_theta(self, shape):
        th = self.fx.t(f"theta{self.nt}").cuda()
        self.nt += 1
        want = [shape] if isinstance(shape, int) else list(shape)
        assert list(th.shape[:len(want)]) == want
        return th


@pytest.mark.parametrize("precision", ["f32", "f16x3"])
def test_eval_boed_on_the_reference_tape(golden, precision):
    """`eval_boed` (utils/eval.py:142-198; stepwise, err_type 'se') against the bounds the REFERENCE's eval_boed returned, on
    the same model, with every random draw of the reference's call replayed in call order (fixture `eval_boed_loc`,
    oracle/make_golden_r4.py: M = 12 outer samples in 2 batches of 6, T = 5, L = 64): get_traces -> argmax designs -> histories
    -> sPCE / sNMC per step -> mean and standard error over the outer samples."""
    from aline_amd.tasks import HiddenLocation
    from aline_amd.utils import eval_boed, get_traces
    fx = golden("eval_boed_loc")
    m = fx.meta
    model, _ = native_model(m["dims"], m["wseed"], precision)
    task = HiddenLocation(n_query_init=m["n_q0"])
    tape = _Tape(fx, task)
    out = eval_boed(model, task, T=m["T"], L=m["L"], M=m["M"], batch_size=m["B"], stepwise=True, err_type="se")
    assert tape.nb == m["n_batch"] and tape.nt == m["n_theta"]               # the product makes the reference's draws, in its order
    for k in ("pce_mean", "pce_err", "nmc_mean", "nmc_err"):
        assert out[k].shape == fx.t(k).shape
        assert maxdiff(out[k], fx.t(k)) < 2e-4, (k, maxdiff(out[k], fx.t(k)))
    # and the histories themselves: the designs the eval-mode model picked, in order of acquisition (eval.py:8-39)
    tape = _Tape(fx, task)
    for i in range(2):
        th0, x, y = get_traces(model, task, m["T"], m["B"], False)
        assert torch.equal(x.cpu(), fx.t(f"trace{i}.x")) and torch.equal(y.cpu(), fx.t(f"trace{i}.y"))
        assert torch.equal(th0.cpu(), fx.t(f"trace{i}.theta0"))
        tape.nt += 1                                                            # (skip the contrastive draw between two traces)


def test_bench_runs_its_rccl_branch_in_a_world_of_one():
    """The `nccl` (= RCCL) branch of bench.py -- init_process_group(device_id=...), the joined-ranks all-reduce, the barriers, the
    max-over-ranks reductions and the flat-buffer gradient all-reduce on a device tensor -- executed on the one GPU of this box
    (ALINE_BENCH_FORCE_DIST=1, a world of one rank), and stdout carries exactly ONE JSON line although RCCL prints its version banner."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(ALINE_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--batch", "64", "--steps", "2", "--warmup", "1",
                          "--train-steps", "1", "--sustain-s", "0", "--prewarm-s", "0", "--no-d256", "--no-d512", "--no-f32",
                          "--no-query-gmm", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["config"]["backend"] == "nccl" and j["config"]["forced_world_of_one"] is True
    assert j["train_step"]["rccl_allreduce_per_step"] == 1.0 and j["train_step"]["collective"].startswith("1 all-reduce")


@pytest.mark.parametrize("B,L,T,K", [(200, 120_000, 31, 1), (20, 300_000, 11, 1), (7, 5_000, 36, 1), (33, 20_000, 6, 2)])
def test_fused_eig_history_equals_the_step_by_step_bounds(B, L, T, K):
    """`aline_eig_location_history` (all T steps of a design history in one pass over the contrastive samples, eig.h) against the
    reference's own structure -- one `EIGStepLoss` step + logsumexp per design (loss/eig.py:174-209, utils/eval.py:64-78) on the step
    kernels -- on the same draw: stepwise sPCE / sNMC of every step.  Shapes: the evaluation protocol's outer batch of 200 and of 20
    (several row groups per workgroup), a history longer than 32 steps, K = 2 sources (the general likelihood loop).  The fixture
    test test_compute_eig_from_history_on_reference_draw holds the same entry point to the reference's bounds."""
    from aline_amd.tasks import HiddenLocation
    from aline_amd.utils import compute_EIG_from_history
    torch.manual_seed(B + T)
    task = HiddenLocation(K=K, n_target_theta=2 * K, device=torch.device("cuda"))
    theta0 = task.sample_theta(B)
    x = torch.rand(B, T, 2, device="cuda")
    y = torch.stack([task.forward(x[:, t], theta0) for t in range(T)], 1)
    thetas = task.sample_theta((L, B))
    pf, nf = compute_EIG_from_history(task, theta0, x, y, L=L, batch_size=B, stepwise=True, thetas=thetas)
    ps, ns = compute_EIG_from_history(task, theta0, x, y, L=L, batch_size=B, stepwise=True, thetas=thetas, fused=False)
    assert pf.shape == ps.shape == (B, T)
    assert torch.isfinite(pf).all() and torch.isfinite(nf).all()
    assert maxdiff(pf, ps.cpu()) < 2e-4 and maxdiff(nf, ns.cpu()) < 2e-4, (maxdiff(pf, ps.cpu()), maxdiff(nf, ns.cpu()))
    p1, n1 = compute_EIG_from_history(task, theta0, x, y, L=L, batch_size=B, stepwise=False, thetas=thetas)
    assert maxdiff(p1, ps[:, -1].cpu()) < 2e-4 and maxdiff(n1, ns[:, -1].cpu()) < 2e-4


@pytest.mark.parametrize("B,L,T", [(40, 200_000, 10), (20, 100_000, 10), (7, 30_000, 16)])
def test_fused_ces_eig_history_equals_the_step_by_step_bounds(B, L, T):
    """`aline_eig_ces_history` (all steps of a CES design history in one pass over the contrastive samples; the per-step arithmetic is
    `eig_ces_step_table_kernel`'s) against one `EIGStepLoss` step + logsumexp per design on the step kernels, same draw: stepwise sPCE /
    sNMC of every step (tasks/ces.py:96-115, :169-210; loss/eig.py:174-209; utils/eval.py:64-78).  The outcomes include censored ones
    (y at eps / 1 - eps: the log-cdf branch).  A history the kernel does not take (T > 16) falls back to the step kernels."""
    from aline_amd.tasks import CESTask
    from aline_amd.utils import compute_EIG_from_history
    torch.manual_seed(B + T)
    dev = torch.device("cuda")
    task = CESTask(device=dev)
    theta0 = task.sample_theta(B)
    x = 100.0 * torch.rand(B, T, 6, device=dev)
    y = torch.stack([task.forward(x[:, t], theta0) for t in range(T)], 1)
    assert bool(((y <= task.epsilon) | (y >= 1 - task.epsilon)).any())          # censored outcomes are in the draw
    thetas = task.sample_theta((L, B))
    pf, nf = compute_EIG_from_history(task, theta0, x, y, L=L, batch_size=B, stepwise=True, thetas=thetas)
    ps, ns = compute_EIG_from_history(task, theta0, x, y, L=L, batch_size=B, stepwise=True, thetas=thetas, fused=False)
    assert pf.shape == ps.shape == (B, T)
    fin = torch.isfinite(ps) & torch.isfinite(ns)
    assert torch.equal(torch.isfinite(pf) & torch.isfinite(nf), fin)
    assert fin.float().mean() > 0.5
    assert float((pf - ps)[fin].abs().max()) < 2e-3 and float((nf - ns)[fin].abs().max()) < 2e-3, (float((pf - ps)[fin].abs().max()), float((nf - ns)[fin].abs().max()))
    assert task.native_eig_history(thetas[:100], x.repeat(1, 2, 1), y.repeat(1, 2, 1)) is None      # 2 T > 16 steps: not taken


def test_range_status_word_survives_graph_replays_with_eager_launches_in_between():
    """The status word of the f16 range guard (first word of the workspace) is cleared by a KERNEL node of the captured rollout
    (aline_hip.hip: clear_words_kernel): a captured hipMemsetAsync was seen to fill its bytes with the arguments of the eager launch
    enqueued right behind a replay (round 3, tools/ws_debug.py; the stand-alone check is tools/probes/graph_memset_repro.hip).  Refresh
    (an eager torch kernel with seed / offset arguments) + replay, unsynchronised, many times: the word and its line stay zero."""
    from aline_amd import Aline, Embedder, Encoder, OutputHead
    from aline_amd.rollout import Rollout
    from aline_amd.tasks import HiddenLocation
    torch.manual_seed(123)
    model = Aline(Embedder(2, 1, 32, 128, 2, "theta"), Encoder(32, 128, 4, 0.0, 3), OutputHead(2, 1, 32, 128)).cuda().set_precision("f16x3").train()
    batch = HiddenLocation(n_query_init=100).sample_batch(128)
    ro = Rollout(model, batch, 8, select="sample").capture()
    for i in range(300):
        ro.refresh_uniform()
        ro.replay()
        if i % 100 == 99:
            torch.cuda.synchronize()
            assert ro.ws[:64].view(torch.int32).abs().sum().item() == 0 and ro.range_status() == 0, i


@pytest.mark.gpu
@pytest.mark.parametrize("d,F,B,T,task", [(256, 1024, 1000, 30, "location"), (512, 128, 256, 30, "psychometric")])
def test_fullsize_training_backward_against_the_exact_fp32_per_op_backward(d, F, B, T, task):
    """The training backward at the two full-size wide shapes of the bench / config lines (d = 256 / F = 1024 / B = 1000 / T = 30 and cfg5:
    d = 512 / B = 256 / T = 30, predefined mask): the default path -- forward recompute on `x3 / x5::layer_save_kernel` over 7 500+
    instances per chunk, scaled-f16 gradient products with producer-emitted scale words, key-row K | V products, matrix-pipe attention
    backward, chunks sized by the 144 GB workspace -- against the same rollout's gradients with every one of those switched off
    (generic recompute in exact fp32, exact-fp32 products, VALU attention backward).  The small-shape tests hold each kernel to the
    reference's autograd; this one holds the launch shapes the numbers are quoted on (whole rounds of row chunks, several chunks of
    different key counts, index lists of 200 000+ key rows).  Tolerances as in the A/B test of the switches (ReLU gates on a knife's edge
    move single units between the two recomputes): nothing beyond 3e-3 of a tensor's max |grad|, four tensors in five within 2e-4."""
    from aline_amd import Aline, Embedder, Encoder, OutputHead, _lib
    from aline_amd.rollout import Rollout
    from aline_amd.train import backward, reinforce_terms
    torch.manual_seed(5)
    dev = torch.device("cuda")
    if task == "location":
        from aline_amd.tasks import HiddenLocation
        model = Aline(Embedder(2, 1, d, F, 2, "theta"), Encoder(d, F, 8, 0.0, 3), OutputHead(2, 1, d, F))
        batch = HiddenLocation(n_query_init=200, device=dev).sample_batch(B)
        emb, mask = "theta", "all"
    else:
        from aline_amd.tasks import PsychometricTask
        model = Aline(Embedder(1, 1, d, F, 4, "theta"), Encoder(d, F, 8, 0.0, 3), OutputHead(1, 1, d, F))
        batch = PsychometricTask(n_query_init=200, n_context_init=1, device=dev).sample_batch(B)
        batch["target_mask"] = torch.tensor([False, False, True, True])
        emb, mask = "theta", "predefined"
    model = model.cuda().set_precision("f16x3").train()
    with torch.no_grad():
        for p in model.parameters():
            p.add_(0.01 * torch.randn_like(p))
    grads = []
    with torch.no_grad():
        ro = Rollout(model, batch, T, select="sample").run()
        assert ro.path in ("x3::layer_kernel", "x5::layer_kernel"), ro.path
        assert ro.range_status() == 0
        terms = reinforce_terms(ro, emb, mask)
        for flags in ([], ["NO_BWD_IMAGE_RECOMPUTE", "NO_BWD_KV_SPARSE", "BWD_GRAD_F32", "BWD_RECOMPUTE_F32", "NO_BWD_ATTN_MFMA"]):
            with _lib.debug(*flags):
                for p in model.parameters():
                    p.grad = None
                backward(model, ro, terms["g_logp"], terms["g_ll"])
                torch.cuda.synchronize()
            grads.append({k: p.grad.clone() for k, p in model.named_parameters()})
    new, ref = grads
    floor = 1e-2 * max(float(g.abs().max()) for g in ref.values())
    errs = {}
    for k in ref:
        assert torch.isfinite(new[k]).all(), k
        errs[k] = float((new[k] - ref[k]).abs().max()) / max(float(ref[k].abs().max()), floor)
    worst = max(errs.items(), key=lambda kv: kv[1])
    assert worst[1] < 3e-3, worst
    close = sum(e < 2e-4 for e in errs.values())
    assert close >= 0.8 * len(errs), (close, len(errs), sorted(errs.items(), key=lambda kv: -kv[1])[:6])

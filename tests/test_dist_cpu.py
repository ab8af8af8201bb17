"""CPU, world_size 2 over gloo: the episode-sharding host logic of the multi-GPU path."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from aline_amd.parallel import aggregate_throughput


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = rank * (total // world), (rank + 1) * (total // world) if rank + 1 < world else total
    # every rank "rolls out" its own episodes; rank 1 is slower
    rate, tmax, units = aggregate_throughput(float(hi - lo) * 6000.0, 1.0 + rank, dist)
    covered = torch.zeros(total)
    covered[lo:hi] = 1
    dist.all_reduce(covered)
    if rank == 0:
        out.put((rate, tmax, units, covered.tolist()))
    dist.destroy_process_group()


def test_shards_cover_batch_once_and_rate_uses_max_time():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    total, world, port = 1001, 2, _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, out)) for r in range(world)]
    [p.start() for p in procs]
    rate, tmax, units, covered = out.get(timeout=120)
    [p.join(60) for p in procs]
    assert all(c == 1.0 for c in covered)               # every episode on exactly one rank
    assert tmax == 2.0 and units == total * 6000.0      # max over ranks, sum over ranks
    assert abs(rate - total * 6000.0 / 2.0) < 1e-6


class _FakeRollout:
    """What aline_amd.train.reinforce_terms reads of a finished rollout, on the CPU."""

    def __init__(self, target_ll, log_prob, n_theta):
        import types
        self.target_ll, self.log_prob = target_ll, log_prob
        self.T, self.B, self.n_t = target_ll.shape
        self.m = types.SimpleNamespace(n_theta=n_theta)
        self.tmask = None

    def nlls(self, embedding_type, mask_type="all"):
        nll = -self.target_ll.mean(-1)
        return nll.t(), nll.t()


def _moments_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from aline_amd.train import reinforce_terms
    g = torch.Generator().manual_seed(5)
    T, B, n_t = 7, 6, 2
    ll, lp = torch.randn(T, world * B, n_t, generator=g), torch.randn(world * B, T, generator=g)
    mine = _FakeRollout(ll[:, rank * B:(rank + 1) * B].contiguous(), lp[rank * B:(rank + 1) * B].contiguous(), 2)
    glob = reinforce_terms(mine, "theta", dist=dist, world=world)              # option (ii): global moments
    loc = reinforce_terms(mine, "theta")                                        # option (i): rank-local moments
    # what the gradient all-reduce does to per-rank quantities: average over ranks
    dl = glob["design_loss"].clone()
    dist.all_reduce(dl)
    dl /= world
    if rank == 0:
        whole = reinforce_terms(_FakeRollout(ll, lp, 2), "theta")              # the single-process step on the concatenated batch
        out.put(dict(R_glob=glob["R"], R_loc=loc["R"], R_whole=whole["R"][:B], g_glob=glob["g_logp"], g_whole=whole["g_logp"][:B],
                     dl=float(dl), dl_whole=float(whole["design_loss"])))
    dist.destroy_process_group()


def test_global_reward_moments_equal_the_single_process_batch():
    """SURVEY 8-e option (ii): with the [3, T - 1] moment all-reduce the z-scored rewards of an N-rank step are those of the
    single-process step on the concatenated batch (train_aline.py:122), and so are the design loss and -- after the gradient
    all-reduce's average over ranks -- its gradient; the rank-local z-score (option i) is not."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    world, port = 2, _free_port()
    procs = [ctx.Process(target=_moments_worker, args=(r, world, port, out)) for r in range(world)]
    [p.start() for p in procs]
    res = out.get(timeout=120)
    [p.join(60) for p in procs]
    assert torch.allclose(res["R_glob"], res["R_whole"], atol=1e-5)
    assert not torch.allclose(res["R_loc"], res["R_whole"], atol=1e-3)
    # g_logp = -alpha R / (B_local (T - 1)) per rank; averaged over `world` ranks it is the whole batch's -alpha R / (B (T - 1))
    assert torch.allclose(res["g_glob"] / world, res["g_whole"], atol=1e-6)
    assert abs(res["dl"] - res["dl_whole"]) < 1e-5


def _rng_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import random
    import numpy as np
    from aline_amd.driver import RankRng, epoch_target_mask
    rec = {}
    for start_epoch in (0, 40):                          # a fresh run, and a run resumed at epoch 40 (same restored generator states)
        torch.manual_seed(123); np.random.seed(123); random.seed(123)
        rng = RankRng(rank, world, start_epoch)
        masks, eps = [], []
        for cfg in (dict(mask_type=["partial"], embedding_type="data", n_target_data=12, n_target_theta=0, n_selected_targets=4),
                    dict(mask_type=["predefined"], embedding_type="theta", n_target_data=0, n_target_theta=4,
                         predefined_masks=[[False, False, True, True], [True, True, False, False]], predefined_mask_weights=[1.0, 1.0])):
            for _ in range(4):
                masks.append(rng.shared_draw(lambda: epoch_target_mask(cfg))[1].tolist())
                eps.append(torch.rand(3).tolist())       # "episodes": the rank's own torch stream
        rec[start_epoch] = (masks, eps)
    out.put((rank, rec))


def test_ranks_draw_the_same_masks_and_their_own_episodes_also_after_resume():
    """`partial` / weighted `predefined` masks consume torch's generator (utils/target_mask.py:18,24), which is
    re-seeded per rank for the episodes -- the driver draws them from the stream all ranks share; a resumed run must not
    replay the episode stream of the original run's first epochs."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    world, port = 2, _free_port()
    procs = [ctx.Process(target=_rng_worker, args=(r, world, port, out)) for r in range(world)]
    [p.start() for p in procs]
    got = dict(out.get(timeout=120) for _ in range(world))
    [p.join(60) for p in procs]
    for start in (0, 40):
        assert got[0][start][0] == got[1][start][0]                     # same masks on both ranks
        assert got[0][start][1] != got[1][start][1]                     # different episodes
    assert len({tuple(map(tuple, got[0][0][0]))}) == 1 and any(m != got[0][0][0][0] for m in got[0][0][0])   # the draws vary
    assert got[0][0][1] != got[0][40][1] and got[1][0][1] != got[1][40][1]      # resumed: a new episode stream per rank


def _grad_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from aline_amd.train import all_reduce_grads
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3))
    for i, p in enumerate(model.parameters()):
        p.grad = torch.full_like(p, float(rank + 1) * (i + 1))
    all_reduce_grads(model, dist, world)          # one flat bucket: sum over ranks / world
    if rank == 0:
        out.put([float(p.grad.mean()) for p in model.parameters()])
    dist.destroy_process_group()


def _flat_grad_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from aline_amd.train import all_reduce_grads
    model = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.ReLU(), torch.nn.Linear(7, 3))
    params = list(model.parameters())
    flat = torch.zeros(sum(p.numel() for p in params))
    off = 0
    for i, p in enumerate(params):                      # the layout train.flat_grads builds: every .grad a view of `flat`
        p.grad = flat[off:off + p.numel()].view_as(p)
        p.grad.fill_(float(rank + 1) * (i + 1))
        off += p.numel()
    all_reduce_grads(model, dist, world, flat=flat)     # ONE collective on the buffer itself, no gather / scatter copies
    if rank == 0:
        out.put([float(p.grad.mean()) for p in params] + [float(p.grad.data_ptr() == flat.data_ptr()) for p in params[:1]])
    dist.destroy_process_group()


def _status_worker(rank, world, port, out):
    """The f16 range status of a training step rides the gradient all-reduce: train.flat_grads puts one slot behind the gradients,
    all_reduce_grads reduces gradients + slot in ONE collective, check_training_range raises on every rank if ANY rank set it."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import aline_amd.train as tr
    model = torch.nn.Linear(3, 2)
    flat, store = None, None
    params = list(model.parameters())
    store = torch.zeros(sum(p.numel() for p in params) + 1)
    flat = store[:-1]
    off = 0
    for p in params:
        p.grad = flat[off:off + p.numel()].view_as(p)
        p.grad.fill_(float(rank + 1))
        off += p.numel()
    model._aline_flat_store = store
    before = tr.ALLREDUCE_CALLS
    store[-1] = 1.0 if rank == 1 else 0.0               # only rank 1 overflowed
    tr.all_reduce_grads(model, dist, world, flat=flat)
    raised = False
    try:
        tr.check_training_range({"range_status": store[-1:]})
    except RuntimeError:
        raised = True
    out.put((rank, raised, float(flat.mean()), tr.ALLREDUCE_CALLS - before))
    dist.barrier()
    dist.destroy_process_group()


def test_range_status_rides_the_gradient_all_reduce_and_stops_every_rank():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    world, port = 2, _free_port()
    procs = [ctx.Process(target=_status_worker, args=(r, world, port, out)) for r in range(world)]
    [p.start() for p in procs]
    res = sorted(out.get(timeout=120) for _ in range(world))
    [p.join(60) for p in procs]
    assert res == [(0, True, 1.5, 1), (1, True, 1.5, 1)]       # both ranks stop; gradients averaged; one collective


def test_flat_buffer_all_reduce_keeps_the_gradient_views():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    world, port = 2, _free_port()
    procs = [ctx.Process(target=_flat_grad_worker, args=(r, world, port, out)) for r in range(world)]
    [p.start() for p in procs]
    res = out.get(timeout=120)
    [p.join(60) for p in procs]
    assert res[:4] == [1.5 * (i + 1) for i in range(4)] and res[4] == 1.0


def test_flat_bucket_gradient_all_reduce_averages_over_ranks():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    world, port = 2, _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, out)) for r in range(world)]
    [p.start() for p in procs]
    means = out.get(timeout=120)
    [p.join(60) for p in procs]
    assert means == [1.5 * (i + 1) for i in range(4)]     # mean of rank values (1, 2) * (i + 1)


def _gather_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from aline_amd.utils.eval import bound_statistics, gather_rows
    # rank r holds outer batches r, r + world, ... of 5 batches of 4 samples each (uneven: 3 vs 2 batches), T = 3
    mine = [i for i in range(5) if i % world == rank]
    pce = torch.cat([torch.arange(4 * 3, dtype=torch.float32).reshape(4, 3) + 100 * i for i in mine])
    got = gather_rows(pce, dist, world)
    stats = bound_statistics(got, -got, "se")
    if rank == 0:
        out.put((got, stats["pce_mean"], stats["nmc_err"]))
    dist.destroy_process_group()


def test_eval_bounds_gather_over_ranks():
    """eval_boed shards the outer batches over the ranks; every rank must end with all of them (rank order) and
    the statistics of the single-process evaluation."""
    from aline_amd.utils.eval import bound_statistics
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    world, port = 2, _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, world, port, out)) for r in range(world)]
    [p.start() for p in procs]
    got, mean, err = out.get(timeout=120)
    [p.join(60) for p in procs]
    blocks = {i: torch.arange(12, dtype=torch.float32).reshape(4, 3) + 100 * i for i in range(5)}
    expect = torch.cat([blocks[i] for i in (0, 2, 4, 1, 3)])            # rank 0's batches, then rank 1's
    assert torch.equal(got, expect)
    ref = bound_statistics(expect, -expect, "se")
    assert torch.allclose(mean, ref["pce_mean"]) and torch.allclose(err, ref["nmc_err"])
    single = bound_statistics(torch.cat([blocks[i] for i in range(5)]), -torch.cat([blocks[i] for i in range(5)]), "se")
    assert torch.allclose(mean, single["pce_mean"]) and torch.allclose(err, single["nmc_err"])   # order-independent


def test_bench_gpus_n_without_launcher_starts_n_ranks():
    """`bench.py --gpus N` called directly (no torchrun environment) must start N ranks itself -- one per GPU through
    torch.distributed.run on 127.0.0.1 -- before touching the GPU, never report a 1-rank number as the N-GPU point; under a
    launcher that started a different number of ranks it must refuse."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["ALINE_BENCH_DRY_SPAWN"] = "1"
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "3", "--warmup", "1"],
                         env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    cmd = json.loads(out.stdout.strip().splitlines()[-1])["spawn"]
    assert "torch.distributed.run" in cmd and "--nproc-per-node=4" in cmd and "127.0.0.1" in cmd
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    env2 = dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4"], env=env2, capture_output=True,
                         text=True, timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE" in (bad.stderr + bad.stdout)

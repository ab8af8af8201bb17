"""CPU, world_size 2 over gloo: the episode-sharding host logic of the multi-GPU path."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from aline_amd.parallel import aggregate_throughput, shard_range


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(total, rank, world)
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


def test_shard_range_properties():
    for total in (0, 1, 7, 1000, 4096):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


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

#!/usr/bin/env python3
"""bench.py -- candidate designs scored / second on the north-star config
(location_finding K=1, batch=1000, T=30, n_query_init=200; BASELINE.json configs[1]).

One "step" = one full T-step acquisition rollout (embedder -> encoder -> head -> design selection ->
context update, T times) of `--batch` synthetic episodes per GPU, inputs resident in HBM.
Multi-GPU: episodes are independent, every rank rolls out its own batch (weak scaling, no
data-path collective in the forward rollout); value = all ranks' designs / max-over-ranks time.

    python bench.py                       # 1 GPU, defaults finish in a few minutes
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# peak constants, pinned in one place (/opt/skills/guides/MI355X_MICROARCH.md, chip-level table)
PEAK_BF16_DENSE_TFLOPS = 2500.0     # dense bf16 MFMA (never the 2:1-sparse 5 PF figure)
PEAK_F32_MFMA_TFLOPS = 157.3        # f32-input MFMA == f32 vector peak
PEAK_HBM_GBS = 8000.0


def algorithmic_flops_per_episode(dx, dy, d, F, H, L, C, n_c0, n_q0, n_td, n_th, n_s, T,
                                  embedding_type="theta", with_query_gmm=True):
    """Forward algorithmic FLOPs of one episode summed over T steps (SURVEY.md 8-d formula)."""
    total = 0.0
    n_t = n_td + n_th
    for t in range(T):
        n_c, n_q = n_c0 + t, n_q0 - t
        N = n_c + n_q + n_t
        tok_x = n_c + n_q + (n_td if embedding_type in ("data", "mix") else 0)
        E = 2 * (dx * F + F * d) * tok_x + 2 * (dy * F + F * d) * n_c
        QKV = 2 * N * d * d + 4 * (n_c + n_s) * d * d
        ATT = 4 * d * ((n_c + n_t) * n_c + n_q * (n_c + n_s))
        REST = 2 * N * d * d + 4 * N * d * F
        ACQ = 2 * n_q * (d * F + F)
        GMM = C * 2 * (d * F + 3 * dy * F) * (n_t + (n_q if with_query_gmm else 0))
        total += E + L * (QKV + ATT + REST) + ACQ + GMM
    return total


def host_cores():
    """Threads this process may really use: min(affinity, cgroup cpu quota)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("ALINE_CPU_THREADS", "16"))))


def build_model(args, device):
    from aline_amd import Aline, Embedder, Encoder, OutputHead
    torch.manual_seed(args.seed)
    model = Aline(Embedder(2, 1, args.d_model, args.d_ff, 2, "theta"),
                  Encoder(args.d_model, args.d_ff, args.heads, 0.0, args.layers),
                  OutputHead(2, 1, args.d_model, args.d_ff, num_components=10))
    return model.to(device).set_precision(args.precision)


def cpu_baseline(args, model):
    """The CPU oracle (pinned to the reference by tests/golden) timed on this host's cores, on a
    bounded sample of the same workload: `cpu_batch` episodes, full T, eval-mode forward."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import aline_oracle as orc
    cores = host_cores()
    torch.set_num_threads(cores)
    log(f"cpu baseline on {cores} threads")
    sd = orc.cast_state_dict(model.state_dict())
    cfg = dict(embedding_type="theta", n_head=args.heads, num_layers=args.layers, num_components=10,
               std_min=1e-4, n_target_theta=2)
    g = torch.Generator().manual_seed(args.seed)

    def make(B):
        x = torch.rand(B, 1 + args.n_query, 2, generator=g)
        th = torch.rand(B, 1, 2, generator=g)
        sig = orc.location_forward_signal(x, th.unsqueeze(1).expand(B, x.shape[1], 1, 2))
        y = sig + 0.5 * torch.randn(sig.shape, generator=g)
        return dict(context_x=x[:, :1], context_y=y[:, :1], query_x=x[:, 1:], query_y=y[:, 1:],
                    target_all=th.reshape(B, 2, 1))

    with torch.no_grad():
        orc.rollout(sd, make(4), cfg, 2, with_query_gmm=True)          # warm-up
        t0 = time.perf_counter()
        orc.rollout(sd, make(16), cfg, args.T, with_query_gmm=True)    # calibration
        cal = time.perf_counter() - t0
        # bounded sample: about 15 s of CPU work, at most --cpu-batch episodes
        B = int(max(16, min(args.cpu_batch, 16 * 15.0 / max(cal, 1e-3))))
        log(f"cpu baseline calibration: 16 episodes in {cal:.2f} s -> timing {B} episodes")
        t0 = time.perf_counter()
        orc.rollout(sd, make(B), cfg, args.T, with_query_gmm=True)
        dt = time.perf_counter() - t0
    return {"value": B * args.T * args.n_query / dt, "unit": "designs/s", "cores": cores,
            "kind": "port",
            "sample": f"{B} episodes x T={args.T} x n_query={args.n_query}, fp32 torch-CPU oracle, "
                      f"eval forward incl. posterior_out_query, {dt:.1f} s"}


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1000, help="episodes per GPU")
    ap.add_argument("--T", type=int, default=30)
    ap.add_argument("--n-query", type=int, default=200)
    ap.add_argument("--d-model", type=int, default=32)
    ap.add_argument("--d-ff", type=int, default=128)
    ap.add_argument("--heads", type=int, default=4)
    ap.add_argument("--layers", type=int, default=3)
    ap.add_argument("--precision", default="bf16x3", choices=["f32", "bf16", "bf16x3"])
    ap.add_argument("--graph", type=int, default=1, help="replay the rollout from one HIP graph")
    ap.add_argument("--cpu-batch", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--seed", type=int, default=123)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=device)

    from aline_amd.rollout import Rollout
    from aline_amd.tasks import HiddenLocation

    log(f"rank {rank}/{world}: building model d={args.d_model} F={args.d_ff} prec={args.precision}")
    model = build_model(args, device)
    model.train()                       # train-time path: designs are sampled (head.py:350-354)
    torch.manual_seed(args.seed + rank)             # every rank rolls out its own episodes
    task = HiddenLocation(n_query_init=args.n_query, device=device)
    batch = task.sample_batch(args.batch)
    ro = Rollout(model, batch, args.T, select="sample", keep_zt=False, keep_posterior=True)
    log("rollout object built; first (eager) rollout ...")
    ro.run()
    torch.cuda.synchronize(device)
    log("eager rollout done")
    if args.graph:
        ro.capture()
        log("graph captured")
    run = ro.replay if args.graph else ro.run

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(args.warmup):
        ro.refresh_uniform()
        run()
    barrier()
    log("warm-up done; timing ...")
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        ro.refresh_uniform()
        run()
    ev1.record()
    barrier()
    dt = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    if dist is not None:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    designs_per_rollout = args.batch * args.T * args.n_query
    exact = args.batch * sum(args.n_query - t for t in range(args.T))
    value = world * designs_per_rollout * args.steps / dt
    fl_ep = algorithmic_flops_per_episode(2, 1, args.d_model, args.d_ff, args.heads, args.layers, 10,
                                          1, args.n_query, 0, 2, 2, args.T, with_query_gmm=False)
    achieved_tflops = fl_ep * args.batch * args.steps / (dev_ms * 1e-3) / 1e12
    peak = PEAK_F32_MFMA_TFLOPS if args.precision == "f32" else PEAK_BF16_DENSE_TFLOPS
    out = {
        "metric": "candidate designs scored/sec (batch x T x n_query), location_finding T=30",
        "value": value, "unit": "designs/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None,
        "dtype": {"f32": "f32", "bf16": "bf16", "bf16x3": "bf16x3(split-bf16 MFMA, fp32 accumulate)"}[args.precision],
        "data": "synthetic",
        "config": {"workload": "location_finding K=1 rollout fwd (embed+encoder+head+select+update), "
                               "train-mode sampling",
                   "batch_per_gpu": args.batch, "T": args.T, "n_query_init": args.n_query,
                   "n_tokens": 1 + args.n_query + 2, "d_model": args.d_model, "d_ff": args.d_ff,
                   "heads": args.heads, "layers": args.layers, "components": 10,
                   "posterior_out_query": "lazy (not computed)", "hip_graph": bool(args.graph),
                   "exact_designs_per_rollout": exact, "parallelism": f"episode-dp{world}"},
        "roofline": {"bound": "mfma", "achieved": achieved_tflops, "peak": peak, "unit": "TFLOP/s",
                     "frac": achieved_tflops / peak, "traffic": None,
                     "kernel": "whole rollout graph (all kernels of T steps; per-kernel split in profiles/)",
                     "algorithmic_flops_per_episode": fl_ep, "device_ms_per_rollout": dev_ms / args.steps},
    }
    log(f"timed region done: {dt / args.steps * 1e3:.2f} ms per rollout")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, model)
        log("cpu baseline done")
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

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


def fused_kernel_flops_per_episode(d, F, L, n_c0, n_q0, n_t, n_s, T):
    """Algorithmic FLOPs of what fused::rollout_f32_kernel computes for one episode: encoder layers +
    acquisition MLP, summed over T steps (embedder and target GMM run in side kernels)."""
    total = 0.0
    for t in range(T):
        n_c, n_q = n_c0 + t, n_q0 - t
        N = n_c + n_q + n_t
        QKV = 2 * N * d * d + 4 * (n_c + n_s) * d * d
        ATT = 4 * d * ((n_c + n_t) * n_c + n_q * (n_c + n_s))
        REST = 2 * N * d * d + 4 * N * d * F
        ACQ = 2 * n_q * (d * F + F)
        total += L * (QKV + ATT + REST) + ACQ
    return total


def fused_kernel_mfma_cycles_per_simd(L, n_c0, n_q0, n_t, n_s, T):
    """MFMA issue cycles one SIMD spends on one episode in fused::rollout_f32_kernel (3 waves of an
    episode share a SIMD).  v_mfma_f32_16x16x4_f32 = 32 cycles, v_mfma_f32_16x16x32_bf16 = 16 cycles
    (MI355X_MICROARCH.md, per-instruction cycle constants).  Per 16-token tile and layer:
      fp32 MFMAs:  Wq 16 + Wo 16 + scores 16 * kt + PV 16 * kt     (kt = 1 while <= 16 keys, else 2)
      split-bf16:  FFN 16 blocks x 6 passes;   acquisition MLP: 8 blocks x 6 passes per tile and step
    plus the K/V pre-pass of the key tiles (2 blocks x 8 fp32 MFMAs per item, 2 * kt items)."""
    cyc = 0.0
    N = n_c0 + n_q0 + n_t
    tiles = (N + 15) // 16
    for t in range(T):
        kt = 2 if (n_c0 + t + n_s) > 16 else 1
        per_tile_layer = (32 + 32 * kt) * 32 + 16 * 6 * 16
        cyc += L * (tiles * per_tile_layer + 2 * kt * 16 * 32) + tiles * 8 * 6 * 16
    return cyc


class HipEvents:
    """A hipEvent_t pair created through libamdhip64 (torch's own HIP runtime) so that the C ABI can
    record them on the launch stream around the dominant kernel."""

    def __init__(self):
        import ctypes as C
        self.C = C
        self.hip = C.CDLL("libamdhip64.so")
        self.a, self.b = C.c_void_p(), C.c_void_p()
        assert self.hip.hipEventCreate(C.byref(self.a)) == 0
        assert self.hip.hipEventCreate(C.byref(self.b)) == 0

    def elapsed_ms(self):
        """None when the library never recorded the pair (the launch took a path without a dominant kernel, e.g.
        ALINE_DISABLE_FUSED=1): the caller then reports the whole graph instead."""
        ms = self.C.c_float()
        if self.hip.hipEventSynchronize(self.b) != 0:
            return None
        if self.hip.hipEventElapsedTime(self.C.byref(ms), self.a, self.b) != 0:
            return None
        return float(ms.value)


def dominant_kernel_ms(ro, device):
    """Launch durations of the path's dominant kernel at EVERY step t of a rollout (dict t -> ms): HIP events recorded by the C ABI on
    the launch stream around that launch (aline_rollout.ev_kernel_start / _stop / _step), one eager rollout per timed step.  The
    roofline figure pairs the SUM of the algorithmic FLOPs of the timed launches with the SUM of their durations, i.e. the average
    launch -- what a rocprofv3 --kernel-trace --stats average of the same command measures (profiles/, cross-checked there).
    Empty when the library never recorded the pair (a path without a dominant kernel)."""
    ev = HipEvents()
    ro.r.ev_kernel_start, ro.r.ev_kernel_stop = ev.a, ev.b
    per = {}
    try:
        for t in range(ro.T):
            ro.r.ev_kernel_step = t + 1
            ro.refresh_uniform()
            ro.run()
            torch.cuda.synchronize(device)
            ms = ev.elapsed_ms()
            if ms is None:
                return {}
            per[t] = ms
    finally:
        ro.r.ev_kernel_start, ro.r.ev_kernel_stop, ro.r.ev_kernel_step = None, None, 0
    return per


def profile_avg_launch_us(csv_name, kernel_prefix):
    """Average launch duration (us) of a kernel in a committed rocprofv3 --kernel-trace --stats summary (profiles/<csv_name>), or None."""
    import csv
    try:
        for r in csv.DictReader(open(os.path.join(ROOT, "profiles", csv_name))):
            if r["Name"].replace("void ", "").startswith(kernel_prefix):
                return float(r["AverageNs"]) / 1e3
    except Exception:
        pass
    return None


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
    """The CPU oracle (pinned to the reference by tests/golden) timed on this host's cores, on bounded samples of the same
    workload (full T): (a) eval-mode forward, the work `value` times (posterior_out_query lazy); (b) the same with the query GMM
    of every step, beside `value_with_query_gmm`; (c) train mode -- forward with autograd + loss.backward() of
    train_aline.py:80-132 --, beside `train_step`."""
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

    def timed(fn, budget_s, cap):
        """calibrate on 16 episodes, then time about `budget_s` seconds of CPU work (at most `cap` episodes)"""
        t0 = time.perf_counter()
        fn(16)
        cal = time.perf_counter() - t0
        B = int(max(16, min(cap, 16 * budget_s / max(cal, 1e-3))))
        t0 = time.perf_counter()
        fn(B)
        return B, time.perf_counter() - t0

    def fwd(B, with_q=False):
        with torch.no_grad():
            orc.rollout(sd, make(B), cfg, args.T, with_query_gmm=with_q)

    def train(B):
        p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        forced = torch.stack([torch.randint(0, args.n_query - t, (B,), generator=g) for t in range(args.T)], 1)
        res = orc.rollout(p, make(B), cfg, args.T, forced_idx=forced)
        _, design, predict = orc.reinforce_losses(torch.stack(res["log_prob"], 1), res["nll_q"], res["nll"])
        (design + predict).backward()

    with torch.no_grad():
        orc.rollout(sd, make(4), cfg, 2, with_query_gmm=False)         # warm-up
    B, dt = timed(fwd, 12.0, args.cpu_batch)
    out = {"value": B * args.T * args.n_query / dt, "unit": "designs/s", "cores": cores, "kind": "port",
           "sample": f"{B} episodes x T={args.T} x n_query={args.n_query}, fp32 torch-CPU oracle, "
                     f"same work as the GPU path (posterior_out_query lazy), {dt:.1f} s"}
    log(f"cpu baseline forward: {B} episodes in {dt:.1f} s")
    if not args.no_query_gmm:
        Bq, dq = timed(lambda n: fwd(n, True), 6.0, args.cpu_batch)
        out["with_query_gmm"] = {"value": Bq * args.T * args.n_query / dq, "unit": "designs/s",
                                 "sample": f"{Bq} episodes, posterior_out_query of every step computed (model/head.py:366), {dq:.1f} s"}
    if args.train_steps > 0:
        Bt, dtt = timed(train, 8.0, args.cpu_batch)
        out["train_step"] = {"value": Bt * args.T * args.n_query / dtt, "unit": "designs/s",
                             "sample": f"{Bt} episodes: teacher-forced rollout under autograd + REINFORCE losses + loss.backward() "
                                       f"(train_aline.py:80-132), no optimiser step, {dtt:.1f} s"}
        log(f"cpu baseline train mode: {Bt} episodes in {dtt:.1f} s")
    return out


def x3_layer_flops(d, F, n_c, n_q, n_t, n_s):
    """Algorithmic FLOPs of ONE x3::layer_kernel launch per episode (one encoder layer of one step, SURVEY 8-d terms): Q
    projection of all N rows, masked attention core, out-projection, FFN.  (K / V of the key rows: x3::kv_kernel.)"""
    N = n_c + n_q + n_t
    return 2 * N * d * d + 4 * d * ((n_c + n_t) * n_c + n_q * (n_c + n_s)) + 2 * N * d * d + 4 * N * d * F


def measure_d256(args, device, batch, precision, d=256, F=1024, train=True):
    """Sub-measurement (not `value`): the same workload on a matrix-core-bound model width: d = 256 / F = 1024 / H = 8 (the north
    star's d_model >= 256 variant) or d = 512 / F = 128 / H = 8 (the width and literal FFN size of BASELINE configs[4], psychometric).
    precision f16x3 = the reference-precision x3 / x5 path (every product a 3-term f16 split, posterior NLL within 1e-4 of the
    reference: tests/test_hip_parity.py, tests/test_r2_gpu.py, tests/test_x5_gpu.py, tests/test_r4_gpu.py: oracle slices of this very launch shape); bf16 = the generic pipeline on the
    single-pass bf16 GEMM policy (throughput mode, NLL error ~1e-2; the bf16 `wide` kernels of rounds 1-3 were removed in round 4)."""
    from aline_amd import Aline, Embedder, Encoder, OutputHead
    from aline_amd.rollout import Rollout
    H, L = 8, args.layers
    torch.manual_seed(args.seed)
    model = Aline(Embedder(2, 1, d, F, 2, "theta"), Encoder(d, F, H, 0.0, L), OutputHead(2, 1, d, F, num_components=10))
    model = model.to(device).set_precision(precision).train()
    ro = Rollout(model, batch, args.T, select="sample", keep_zt=False, keep_posterior=True)
    ro.run(); torch.cuda.synchronize(device)
    ro.capture()
    ro.refresh_uniform(); ro.replay(); torch.cuda.synchronize(device)
    steps = 3
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        ro.refresh_uniform(); ro.replay()
    e1.record(); torch.cuda.synchronize(device)
    ms = e0.elapsed_time(e1) / steps
    per_step = dominant_kernel_ms(ro, device)                      # the last layer's launch of every step t
    kernel_ms = sum(per_step.values()) / len(per_step) if per_step else None
    fl_ep = algorithmic_flops_per_episode(2, 1, d, F, H, L, 10, 1, args.n_query, 0, 2, 2, args.T, with_query_gmm=False)
    whole = fl_ep * args.batch / (ms * 1e-3) / 1e12
    out = {"model": f"d={d} F={F} H={H} L={L}", "precision": precision, "ms_per_rollout": ms,
           "value": args.batch * args.T * args.n_query / (ms * 1e-3), "unit": "designs/s",
           "whole_rollout": {"achieved": whole, "peak": PEAK_BF16_DENSE_TFLOPS, "unit": "TFLOP/s", "frac": whole / PEAK_BF16_DENSE_TFLOPS,
                             "algorithmic_flops_per_rollout": fl_ep * args.batch}}
    if kernel_ms:
        passes = 3 if precision == "f16x3" else 1
        kname = ro.kernel_name
        if precision == "f16x3":           # average launch: the timed launches' algorithmic FLOPs / their durations
            per_launch = sum(x3_layer_flops(d, F, 1 + t, args.n_query - t, 2, 2) for t in per_step) / len(per_step) * args.batch
        else:
            fl_all = fused_kernel_flops_per_episode(d, F, L, 1, args.n_query, 2, 2, args.T)
            per_launch = fl_all / args.T * args.batch
        ach = per_launch / (kernel_ms * 1e-3) / 1e12
        out["roofline"] = {"bound": "mfma", "kernel": kname, "kernel_ms_per_launch": kernel_ms,
                           "algorithmic_flops_per_launch": per_launch, "achieved": ach, "peak": PEAK_BF16_DENSE_TFLOPS,
                           "unit": "TFLOP/s", "frac": ach / PEAK_BF16_DENSE_TFLOPS, "mfma_passes_per_product": passes,
                           "instruction_mix_peak": PEAK_BF16_DENSE_TFLOPS / passes,
                           "frac_vs_instruction_mix_peak": ach * passes / PEAK_BF16_DENSE_TFLOPS,
                           "timed_launches": len(per_step),
                           **profile_traffic(("x3_f16x3_d256_pmc_traffic.json" if (d, F) == (256, 1024) else "x5_f16x3_d512_pmc_traffic.json" if (d, F) == (512, 128) else None)
                                             if precision == "f16x3" else None, args.batch == 1000 and args.T == 30 and args.n_query == 200),
                           "peak_note": "dense f16/bf16 MFMA peak (MI355X_MICROARCH.md).  A reference-precision product costs "
                                        "3 MFMA passes (hi*hi + hi*lo + lo*hi), so the pipe can deliver at most peak / 3 of "
                                        "algorithmic FLOP/s in this mode: instruction_mix_peak; frac_vs_instruction_mix_peak is "
                                        "the matrix-pipe utilisation (PMC SQ_VALU_MFMA_BUSY_CYCLES agrees: profiles/)"}
    out["f16_range_status"] = ro.range_status()
    if train and args.train_steps > 0 and precision == "f16x3":
        # the training step of the same model (per-op exact-fp32 backward at this width)
        from aline_amd.train import train_step
        try:
            opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
            train_step(model, batch, args.T, optimizer=opt)
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            train_step(model, batch, args.T, optimizer=opt)
            torch.cuda.synchronize(device)
            out["train_step"] = {"ms_per_step": (time.perf_counter() - t0) * 1e3, "steps": 1,
                                 "value": args.batch * args.T * args.n_query / (time.perf_counter() - t0), "unit": "designs/s"}
        except Exception as e:
            out["train_step"] = {"error": repr(e)}
    return out


def profile_traffic(fname, same_shape):
    """roofline.traffic of a leg: HBM bytes per launch of its dominant kernel from the rocprofv3 PMC passes committed under
    profiles/ (FETCH_SIZE x 2 + WRITE_SIZE, MI355X_MICROARCH.md; collected by tools/pmc_*.sh on the same kernel and shape,
    not re-measured inside bench.py: PMC collection serialises the kernels).  null when the shape differs or no profile exists."""
    if fname and same_shape:
        for rnd in ("r04_", "r03_", "r02_", ""):       # the newest round's collection (tools/collect_r04.sh writes the r04_ files) first
            cand = fname if fname.startswith(("r0", "r1")) else rnd + fname
            try:
                tr = json.load(open(os.path.join(ROOT, "profiles", cand)))
                return {"traffic": tr["hbm_bytes_per_launch"],
                        "traffic_source": f"rocprofv3 PMC passes of this kernel at this shape, average launch (profiles/{cand}, written by the "
                                          f"collection script that also ran this bench command: tools/collect_r04.sh); PMC collection serialises "
                                          f"the kernels, so it is a separate pass, not this run"}
            except Exception:
                pass
    return {"traffic": None, "traffic_source": "no PMC profile of this kernel at this shape under profiles/"}


def measure_eig(device, n=10):
    """Sub-measurement (not `value`): the sequential-EIG step kernels (loss/eig.py:174-209 through the task likelihoods,
    tasks/location_finding.py:110-164, tasks/ces.py:169-210) at the README evaluation sizes -- location finding L = 1e6 contrastive
    samples x B = 200, CES L = 1e6 x B = 20 (BASELINE configs[3]).  HBM-bound: algorithmic bytes per (l, b) and step =
    theta read (dim_theta x 4 B) + the running log-likelihood sum S read + written (8 B); the logsumexp finalisation reads S (4 B)."""
    from aline_amd.loss.eig import EIGStepLoss
    from aline_amd.tasks import CESTask, HiddenLocation
    out = {"peak": PEAK_HBM_GBS, "unit": "GB/s", "bound": "hbm"}
    for name, task, L, B, dth in (("location", HiddenLocation(device=device), 1_000_000, 200, 2), ("ces", CESTask(device=device), 1_000_000, 20, 5)):
        torch.manual_seed(0)
        if name == "location":
            theta = torch.rand(L + 1, B, 1, 2, device=device)
            xi, y = torch.rand(B, 2, device=device), torch.randn(B, 1, device=device)
        else:
            theta = torch.stack([0.01 + 0.99 * torch.rand(L + 1, B, device=device), *(torch.rand(3, L + 1, B, device=device) / 3 + 0.1),
                                 torch.randn(L + 1, B, device=device)], -1).contiguous()
            xi, y = torch.rand(B, 6, device=device) * 100, torch.rand(B, 1, device=device) * 0.9 + 0.05
        crit = EIGStepLoss(L, B, task, device=device)
        for _ in range(2):
            crit.step(y, xi, theta)
        torch.cuda.synchronize(device)
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        for _ in range(n):
            crit.step(y, xi, theta)
        e1.record()
        for _ in range(n):
            crit.forward(y, xi, theta)
        e2.record()
        torch.cuda.synchronize(device)
        dt, dtf = e0.elapsed_time(e1) / n * 1e-3, max(e1.elapsed_time(e2) - e0.elapsed_time(e1), 1e-6) / n * 1e-3
        step_bytes = (L + 1) * B * (dth * 4 + 8)
        out[name] = {"L": L, "B": B, "step_ms": dt * 1e3, "algorithmic_bytes_per_step": step_bytes, "achieved": step_bytes / dt / 1e9,
                     "frac": step_bytes / dt / 1e9 / PEAK_HBM_GBS, "finalize_ms": dtf * 1e3, "finalize_GBps": (L + 1) * B * 4 / dtf / 1e9}
        if name == "location":
            # the whole history of T = 30 designs: T step + logsumexp launches (the reference's structure) against ONE pass over theta
            # (aline_eig_location_history, round 4).  Algorithmic bytes of the fused pass: theta once (8 B per (l, b)).
            from aline_amd.utils import compute_EIG_from_history
            T = 30
            xs, ys = torch.rand(B, T, 2, device=device), torch.randn(B, T, 1, device=device)
            th0, thl = theta[0], theta[1:]
            res = {}
            for fused in (True, False):
                compute_EIG_from_history(task, th0, xs, ys, L=L, batch_size=B, stepwise=True, thetas=thl, fused=fused)
                torch.cuda.synchronize(device)
                f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                f0.record()
                for _ in range(3):
                    compute_EIG_from_history(task, th0, xs, ys, L=L, batch_size=B, stepwise=True, thetas=thl, fused=fused)
                f1.record()
                torch.cuda.synchronize(device)
                res[fused] = f0.elapsed_time(f1) / 3
            out["location_history"] = {"T": T, "L": L, "B": B, "fused_ms": res[True], "stepwise_ms": res[False], "speedup": res[False] / res[True],
                                       "fused_algorithmic_bytes": (L + 1) * B * 8, "fused_GBps": (L + 1) * B * 8 / (res[True] * 1e-3) / 1e9,
                                       "note": "fused: one pass over theta for all T steps (bound: transcendental units, not HBM); stepwise: "
                                               "T x (step kernel + streaming logsumexp), 20 B per (l, b) and step"}
        if name == "ces":
            # the CES history of T = 10 designs (BASELINE.json configs[3]): T step + logsumexp launches against ONE pass over theta
            # (aline_eig_ces_history, round 4): theta once, 20 B per (l, b)
            from aline_amd.utils import compute_EIG_from_history
            T = 10
            th0, thl = theta[0], theta[1:]
            xs = 100.0 * torch.rand(B, T, 6, device=device)
            ys = torch.stack([task.forward(xs[:, t], th0) for t in range(T)], 1)
            res = {}
            for fused in (True, False):
                compute_EIG_from_history(task, th0, xs, ys, L=L, batch_size=B, stepwise=True, thetas=thl, fused=fused)
                torch.cuda.synchronize(device)
                f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                f0.record()
                for _ in range(5):
                    compute_EIG_from_history(task, th0, xs, ys, L=L, batch_size=B, stepwise=True, thetas=thl, fused=fused)
                f1.record()
                torch.cuda.synchronize(device)
                res[fused] = f0.elapsed_time(f1) / 5
            out["ces_history"] = {"T": T, "L": L, "B": B, "fused_ms": res[True], "stepwise_ms": res[False], "speedup": res[False] / res[True],
                                  "fused_algorithmic_bytes": (L + 1) * B * 20, "fused_GBps": (L + 1) * B * 20 / (res[True] * 1e-3) / 1e9}
        del theta, crit
        torch.cuda.empty_cache()
    return out


def measure_alt(args, device, batch, precision):
    """The same workload under another arithmetic mode (sub-measurement, not `value`): f32 = exact fp32 MFMA attention /
    projections + 6-pass split-bf16 FFN, the whole rollout in one launch (fused_rollout.h)."""
    from aline_amd.rollout import Rollout
    import copy
    a2 = copy.copy(args)
    a2.precision = precision
    model = build_model(a2, device).train()
    ro = Rollout(model, batch, args.T, select="sample", keep_zt=False, keep_posterior=True)
    ro.run()
    torch.cuda.synchronize(device)
    ro.capture()
    ro.refresh_uniform(); ro.replay()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ro.refresh_uniform()
        ro.replay()
    torch.cuda.synchronize(device)
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    return {"precision": precision, "path": ro.path, "ms_per_rollout": ms,
            "value": args.batch * args.T * args.n_query / (ms * 1e-3), "unit": "designs/s"}


def rollout_ms_with_query_gmm(args, model, batch, device):
    """The same rollout with posterior_out_query (model/head.py:366: the C GMM heads on the candidate rows) of all T steps
    computed as well (`aline_rollout.postq_*`): ms per rollout, replayed from one HIP graph like the timed region.  The
    product computes it lazily in the step API (no caller in train_aline.py / utils/eval.py reads it) and `value` does not
    include it; SURVEY 8-d asks for the figure with and without."""
    from aline_amd.rollout import Rollout
    ro = Rollout(model, batch, args.T, select="sample", keep_zt=False, keep_posterior=True, keep_query_posterior=True)
    ro.run()
    torch.cuda.synchronize(device)
    ro.capture()
    ro.refresh_uniform(); ro.replay()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ro.refresh_uniform()
        ro.replay()
    torch.cuda.synchronize(device)
    return (time.perf_counter() - t0) / args.steps * 1e3, ro.path


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
    ap.add_argument("--precision", default="f16x3", choices=["f32", "bf16", "bf16x3", "f16x3"],
                    help="arithmetic of the matrix products: f16x3 (default) and f32 are the reference-precision modes")
    ap.add_argument("--no-f32", action="store_true", help="skip the f32 (fused fp32-MFMA kernel) sub-measurement (N = 1 only)")
    ap.add_argument("--graph", type=int, default=1, help="replay the rollout from one HIP graph")
    ap.add_argument("--cpu-batch", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--seed", type=int, default=123)
    ap.add_argument("--train-steps", type=int, default=5,
                    help="also time K full training steps (rollout + backward + all-reduce + AdamW); 0 = skip")
    ap.add_argument("--sustain-s", type=float, default=2.5,
                    help="after the timed steps, replay the rollout back to back for this many seconds (sustained_ms_per_step)")
    ap.add_argument("--prewarm-s", type=float, default=1.0,
                    help="untimed device warm-up in front of the W warm-up steps: the rollout replayed back to back for this many seconds, "
                         "so that the K timed steps run at the clock / cache state of a job in progress (0 = none)")
    ap.add_argument("--no-d256", action="store_true", help="skip the d_model = 256 sub-measurement (N = 1 only)")
    ap.add_argument("--no-d512", action="store_true", help="skip the d_model = 512 / F = 128 sub-measurement (N = 1 only)")
    ap.add_argument("--no-eig", action="store_true", help="skip the sequential-EIG kernels' HBM-roofline sub-measurement (N = 1 only)")
    ap.add_argument("--no-query-gmm", action="store_true", help="skip the value_with_query_gmm figure")
    ap.add_argument("--d256-precs", default="f16x3", help="arithmetic modes of the d_model = 256 sub-measurement (comma separated)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `bench.py --gpus N` without a launcher: start the N ranks here, BEFORE anything touches the GPU (a process that
        # has initialised HIP must not be replaced), one per GPU over RCCL, and hand their exit code on
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        log(f"--gpus {args.gpus} without a launcher: starting {args.gpus} ranks: {' '.join(cmd)}")
        if os.environ.get("ALINE_BENCH_DRY_SPAWN"):          # (tests: show the launch line, start nothing)
            print(json.dumps({"spawn": cmd}))
            raise SystemExit(0)
        raise SystemExit(subprocess.call(cmd))

    # ONE JSON line on stdout is the contract, and libraries write there too (RCCL prints a version banner at communicator creation,
    # gloo its "connected to N peer ranks" lines: seen in the round-4 rehearsals): from here on file descriptor 1 is stderr, and the
    # result line goes to the descriptor the process was started with.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    from aline_amd.parallel import aggregate_throughput, world_info
    rank, local_rank, world = world_info()
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}: the launcher started a different number of ranks")
    # ALINE_BENCH_BACKEND=gloo: rehearsal of the N > 1 code path on a box with fewer GPUs than ranks (the ranks then share
    # the visible cards round-robin and the collectives go through the host); the driver's runs use nccl (= RCCL over xGMI)
    backend = os.environ.get("ALINE_BENCH_BACKEND", "nccl")
    device = torch.device("cuda", local_rank % torch.cuda.device_count() if backend == "gloo" else local_rank)
    torch.cuda.set_device(device)
    dist = None
    # ALINE_BENCH_FORCE_DIST=1 (N = 1): initialise the process group, run the joined-ranks all-reduce, the barriers, the max-over-ranks
    # reductions and the flat-buffer gradient all-reduce in a world of ONE rank -- the single-GPU rehearsal of the RCCL code path
    # (the nccl branch had never executed anywhere); results equal the plain N = 1 run.
    force_dist = os.environ.get("ALINE_BENCH_FORCE_DIST") == "1"
    if world > 1 or force_dist:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:          # (forced world of one without a launcher)
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(sk.getsockname()[1]), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        if backend == "gloo":
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
        joined = torch.ones(1, device=device)
        dist.all_reduce(joined)
        if int(joined.item()) != args.gpus or dist.get_world_size() != args.gpus:
            raise SystemExit(f"{int(joined.item())} ranks joined, --gpus {args.gpus} asked")

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

    if os.environ.get("ALINE_DUMP_MAPS"):      # (diagnostic: where every library of this process is mapped, before the timed loop)
        with open(os.environ["ALINE_DUMP_MAPS"], "w") as f:
            f.write(open("/proc/self/maps").read())
    if args.prewarm_s > 0:              # steady state first: the first rollouts after the capture run ~9 % slower than a job in progress
        tw = time.perf_counter()
        while time.perf_counter() - tw < args.prewarm_s:
            for _ in range(20):
                ro.refresh_uniform()
                run()
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
    designs_per_rollout = args.batch * args.T * args.n_query
    # whole job = the designs of all ranks / the slowest rank's time
    value, dt, _ = aggregate_throughput(float(designs_per_rollout * args.steps), dt, dist, device, always=force_dist)

    # sustained leg: the same replay back to back for >= --sustain-s seconds (so that SMI sampling sees the GPU busy)
    sustained_ms = None
    if args.sustain_s > 0:
        n_sus = max(args.steps, int(args.sustain_s / max(dt / args.steps, 1e-4)) + 1)
        barrier()
        t1 = time.perf_counter()
        for _ in range(n_sus):
            ro.refresh_uniform()
            run()
        barrier()
        _, sdt, _ = aggregate_throughput(1.0, time.perf_counter() - t1, dist, device, always=force_dist)
        sustained_ms = sdt / n_sus * 1e3
        log(f"sustained leg: {n_sus} rollouts in {sdt:.2f} s = {sustained_ms:.3f} ms each")

    exact = args.batch * sum(args.n_query - t for t in range(args.T))
    range_status = ro.range_status()           # f16 range guard: 0 = every F16X3 operand of the timed rollouts stayed in range
    # dominant kernel: HIP events recorded by the C ABI on the launch stream around the fused rollout
    # kernel, in an eager leg of the same process right after the timed region (same inputs, same
    # launches; the graph replays above launch exactly this kernel)
    path = ro.path
    per_step = dominant_kernel_ms(ro, device) if path != "generic pipeline" else {}
    if path == "fused::rollout_f32_kernel" and per_step:      # (one launch per rollout: every "step" timed the same launch)
        per_step = {0: sum(per_step.values()) / len(per_step)}
    kernel_ms = sum(per_step.values()) / len(per_step) if per_step else 0.0
    fl_ep = algorithmic_flops_per_episode(2, 1, args.d_model, args.d_ff, args.heads, args.layers, 10,
                                          1, args.n_query, 0, 2, 2, args.T, with_query_gmm=False)
    fused = path == "fused::rollout_f32_kernel" and kernel_ms > 0.0
    x3 = path == "x3::layer_kernel" and kernel_ms > 0.0
    s3 = path == "s3::step_kernel" and kernel_ms > 0.0
    extra = {}
    if fused:
        fl_k = fused_kernel_flops_per_episode(args.d_model, args.d_ff, args.layers, 1, args.n_query, 2, 2, args.T)
        achieved_tflops = fl_k * args.batch / (kernel_ms * 1e-3) / 1e12
        kname, per_launch = ro.kernel_name, fl_k * args.batch
        # The kernel issues two MFMA kinds (fp32 16x16x4 for attention / projections, split-bf16 x6 for
        # the FFN and the acquisition MLP), so its matrix-pipe roofline is the time its own MFMA stream
        # needs at 100 % issue on all 1024 SIMDs (one episode per SIMD, ceil(B / 1024) rounds).
        cyc = fused_kernel_mfma_cycles_per_simd(args.layers, 1, args.n_query, 2, 2, args.T)
        rounds = -(-args.batch // 1024)
        roof_ms = cyc * rounds / 2.4e9 * 1e3
        peak = PEAK_F32_MFMA_TFLOPS
        mix_peak = fl_k * args.batch / (roof_ms * 1e-3) / 1e12
        extra = {"mfma_issue_roofline_ms": roof_ms, "instruction_mix_peak": mix_peak,
                 "frac_vs_instruction_mix_peak": achieved_tflops / mix_peak,
                 "peak_note": "peak = dense fp32 MFMA peak (MI355X_MICROARCH.md) for the fp32 arithmetic this path "
                              "computes in.  62 % of the multiply-adds run as exact 3-way split-bf16 on the bf16 pipe "
                              "(6 passes x 16 cycles instead of 8 x 32), so the tighter bound is the kernel's own "
                              "MFMA issue time at 2.4 GHz on 1024 SIMDs: instruction_mix_peak / "
                              "frac_vs_instruction_mix_peak"}
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "r01_fused_f32_d32_pmc_traffic.json")))
            traffic = tr["hbm_bytes_per_launch"] if args.batch == 1000 and args.T == 30 else None
            extra["traffic_source"] = "rocprofv3 PMC (profiles/r01_fused_f32_d32_pmc_traffic.json), not re-measured in this run"
        except Exception:
            traffic = None
    elif x3:
        # dominant kernel of the x3 path: x3::layer_kernel -- one launch = one encoder layer of one step for all B episodes
        # (L * T launches per rollout); the events bracket the last layer of every step in turn: average launch
        per_launch = sum(x3_layer_flops(args.d_model, args.d_ff, 1 + t, args.n_query - t, 2, 2) for t in per_step) / len(per_step) * args.batch
        achieved_tflops = per_launch / (kernel_ms * 1e-3) / 1e12
        kname, peak = ro.kernel_name, PEAK_BF16_DENSE_TFLOPS
        tr = profile_traffic("x3_f16x3_d256_pmc_traffic.json", args.batch == 1000 and args.T == 30 and args.n_query == 200 and args.d_ff == 1024)
        traffic = tr["traffic"]
        extra = {"traffic_source": tr["traffic_source"], "launches_per_rollout": args.T * args.layers, "mfma_passes_per_product": 3,
                 "instruction_mix_peak": PEAK_BF16_DENSE_TFLOPS / 3,
                 "frac_vs_instruction_mix_peak": achieved_tflops * 3 / PEAK_BF16_DENSE_TFLOPS,
                 "whole_rollout_tflops": fl_ep * args.batch * args.steps / (dev_ms * 1e-3) / 1e12,
                 "peak_note": "dense f16 MFMA peak (MI355X_MICROARCH.md).  Every product is a 3-term f16 split (reference "
                              "precision), i.e. 3 MFMA passes per algorithmic multiply-add: the pipe can deliver at most "
                              "peak / 3 in this mode (instruction_mix_peak); frac_vs_instruction_mix_peak = matrix-pipe utilisation"}
    elif s3:
        # dominant kernel of the s3 path: s3::step_kernel -- one launch = every encoder layer + the acquisition logits of ONE
        # design step for all B episodes (T launches per rollout); the events bracket the launch of every step in turn, and the
        # figure pairs the AVERAGE launch's algorithmic FLOPs (all T launches / T) with the average of the T durations
        fl_all = fused_kernel_flops_per_episode(args.d_model, args.d_ff, args.layers, 1, args.n_query, 2, 2, args.T)
        per_launch = fl_all / args.T * args.batch
        achieved_tflops = per_launch / (kernel_ms * 1e-3) / 1e12
        kname, peak, traffic = ro.kernel_name, PEAK_BF16_DENSE_TFLOPS, None
        prof_us = profile_avg_launch_us("r04_bench_kernel_stats.csv", "s3::step_kernel") if args.batch == 1000 and args.T == 30 and args.n_query == 200 else None
        extra = {"launches_per_rollout": args.T, "timed_launches": len(per_step), "mfma_passes_per_product": 3,
                 "kernel_us_first_last_step": [per_step[0] * 1e3, per_step[args.T - 1] * 1e3],
                 # the same figure from the committed rocprofv3 --kernel-trace --stats summary of this command (graph launches: no
                 # eager-dispatch gap inside the event pair), when one exists for this shape
                 "profile_avg_launch_us": prof_us,
                 "frac_from_profile_avg": (per_launch / (prof_us * 1e-6) / 1e12 / PEAK_BF16_DENSE_TFLOPS) if prof_us else None,
                 "instruction_mix_peak": PEAK_BF16_DENSE_TFLOPS / 3,
                 "frac_vs_instruction_mix_peak": achieved_tflops * 3 / PEAK_BF16_DENSE_TFLOPS,
                 "whole_rollout_tflops": fl_ep * args.batch * args.steps / (dev_ms * 1e-3) / 1e12,
                 "achieved_vs_fp32_mfma_peak": achieved_tflops / PEAK_F32_MFMA_TFLOPS,
                 "kernel_scope": "since round 4 the launch also holds the design selection of its episodes (one wave per episode behind "
                                 "the last layer's barrier, ~2.5 us of the launch; before: acq_select_wave_kernel, 5.5 us per step on "
                                 "its own): the rollout got 2.5 % faster, this kernel-level fraction ~3 % lower for the same FLOPs "
                                 "(profiles/r04_s3_select_in_kernel.txt)",
                 "limiting_resource": "vector issue, not the matrix pipe: at d = 32 a token tile needs ~900 vector instructions per "
                                      "layer (f16 hi/lo splits of every activation, LayerNorm, softmax) beside ~100 MFMAs; PMC of "
                                      "this launch (profiles/r04_s3_f16x3_d32_pmc_summary.txt): VALU busy 57 - 70 %, matrix pipe busy 27 - 33 %",
                 "peak_note": "dense f16 MFMA peak (MI355X_MICROARCH.md).  Every product is a 3-term f16 split (reference "
                              "precision): the pipe can deliver at most peak / 3 in this mode (instruction_mix_peak).  "
                              "achieved_vs_fp32_mfma_peak: the same fp32-grade FLOP rate against the 157 TFLOP/s of the fp32 "
                              "matrix pipe, the roof of round 1's kernel for this workload"}
        tr = profile_traffic("s3_f16x3_d32_pmc_traffic.json", args.batch == 1000 and args.T == 30 and args.n_query == 200)
        traffic = tr["traffic"]
        extra["traffic_source"] = tr["traffic_source"]
    else:   # generic pipeline: many kernels per step; report the whole graph as a lower bound
        achieved_tflops = fl_ep * args.batch * args.steps / (dev_ms * 1e-3) / 1e12
        kname, per_launch, kernel_ms = "whole rollout graph (generic pipeline, all kernels)", fl_ep * args.batch, dev_ms / args.steps
        peak = PEAK_F32_MFMA_TFLOPS if args.precision == "f32" else PEAK_BF16_DENSE_TFLOPS
        traffic = None
    out = {
        "metric": "candidate designs scored/sec (batch x T x n_query), location_finding T=30",
        "value": value, "unit": "designs/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "ranks": world, "sustained_ms_per_step": sustained_ms,
        "dtype": {"f32": "f32 (fp32 MFMA; FFN/acquisition products as exact 3-way split-bf16, fp32 accumulate)",
                  "bf16": "bf16", "bf16x3": "bf16x3(split-bf16 MFMA, fp32 accumulate)", "f16x3": "f32-grade (3-term split-f16 MFMA, fp32 accumulate)"}[args.precision],
        "data": "synthetic",
        "config": {"workload": "location_finding K=1, batch=1000, T=30, n_query_init=200: T-step rollout "
                               "forward (embed + encoder + heads + design sampling + context update + "
                               "GMM log-likelihood)",
                   "batch_per_gpu": args.batch, "T": args.T, "n_query_init": args.n_query,
                   "n_tokens": 1 + args.n_query + 2, "d_model": args.d_model, "d_ff": args.d_ff,
                   "heads": args.heads, "layers": args.layers, "components": 10,
                   "posterior_out_query": "lazy (not computed; same in the CPU baseline)",
                   "hip_graph": bool(args.graph), "exact_designs_per_rollout": exact, "path": path,
                   "f16_range_status": range_status, "backend": backend if (world > 1 or force_dist) else None, "forced_world_of_one": force_dist, "prewarm_s": args.prewarm_s,
                   "precision": args.precision,
                   "parallelism": f"episode-dp{world}"},
        "roofline": {"bound": "mfma", "achieved": achieved_tflops, "peak": peak, "unit": "TFLOP/s",
                     "frac": achieved_tflops / peak, "traffic": traffic, "kernel": kname,
                     "kernel_ms_per_launch": kernel_ms, "algorithmic_flops_per_launch": per_launch,
                     "algorithmic_flops_per_episode_all_kernels": fl_ep,
                     "device_ms_per_rollout": dev_ms / args.steps,
                     **extra},
    }
    log(f"timed region done: {dt / args.steps * 1e3:.2f} ms per rollout")
    if args.train_steps > 0 and args.precision in ("f32", "f16x3"):
        # secondary measurement (not `value`): one optimiser step of train_aline.py:55-152 -- sampled
        # rollout, REINFORCE terms, native backward of all T steps, ONE flat-bucket RCCL all-reduce of the
        # gradients (N > 1), inf-norm clipping, AdamW.
        from aline_amd import train as train_mod
        from aline_amd.train import train_step
        opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
        err = None
        try:
            for _ in range(2):      # warm-up (the first call builds the rollout's graph and the backward workspace)
                train_step(model, batch, args.T, optimizer=opt, dist=dist, world=world, force_collective=force_dist)
        except Exception as e:      # a secondary leg must not take the headline line with it ...
            err = e
            log(f"train step leg failed: {e!r}")
        if dist is not None:
            # ... but with N > 1 a rank that failed has left its peers inside the gradient all-reduce: nothing can be agreed on any
            # more, so the job ends here with a non-zero exit (torchrun tears the other ranks down) instead of hanging in a barrier
            if err is not None:
                raise SystemExit(f"rank {rank}: train step failed: {err!r}")
            ok = torch.ones(1, device=device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if err is not None:
            out["train_step"] = {"error": repr(err)}
        else:
            barrier()
            ar0 = train_mod.ALLREDUCE_CALLS
            t1 = time.perf_counter()
            for _ in range(args.train_steps):
                train_step(model, batch, args.T, optimizer=opt, dist=dist, world=world, force_collective=force_dist)
            barrier()
            tval, tdt, _ = aggregate_throughput(float(designs_per_rollout * args.train_steps), time.perf_counter() - t1, dist, device, always=force_dist)
            train_mod.check_range_async(block=True)
            out["train_step"] = {"value": tval, "unit": "designs/s",
                                 "ms_per_step": tdt / args.train_steps * 1e3, "steps": args.train_steps,
                                 "includes": f"forward rollout ({path}) + fused exact-fp32 backward of all T steps (layer_fwd / tail / attention-block / acquisition-head kernels) + "
                                             "flat-bucket gradient all-reduce (N>1) + inf-norm clip + AdamW",
                                 "collective": "1 all-reduce / optimiser step" if (world > 1 or force_dist) else "none (N=1)",
                                 "rccl_allreduce_calls": train_mod.ALLREDUCE_CALLS - ar0,
                                 "rccl_allreduce_per_step": (train_mod.ALLREDUCE_CALLS - ar0) / args.train_steps}
            log(f"train step: {tdt / args.train_steps * 1e3:.1f} ms")
    if not args.no_query_gmm and args.precision in ("f32", "f16x3"):
        qms, qpath = rollout_ms_with_query_gmm(args, model, batch, device)
        out["value_with_query_gmm"], qsec, _ = aggregate_throughput(float(designs_per_rollout), qms * 1e-3, dist, device, always=force_dist)
        qms = qsec * 1e3
        out["query_gmm"] = {"ms_per_rollout": qms, "path": qpath,
                            "note": "posterior_out_query (model/head.py:366) is lazy in the step API and is NOT part of `value`; this "
                                    "is the same rollout with the C GMM heads also evaluated on the candidate rows of all T steps "
                                    "(aline_rollout.postq_*: [T, B, P, C] by slot), timed like `value` (SURVEY 8-d: with and without)"}
        log(f"rollout with the query GMM of all T steps: {qms:.2f} ms")
    if world == 1 and not args.no_f32 and args.precision != "f32" and args.d_model == 32:
        out["f32"] = measure_alt(args, device, batch, "f32")
        log(f"f32 [{out['f32']['path']}]: {out['f32']['ms_per_rollout']:.2f} ms per rollout")
    if world == 1 and not args.no_d256 and args.d_model != 256:
        out["d256"] = {}
        for prec in args.d256_precs.split(","):
            out["d256"][prec] = measure_d256(args, device, batch, prec)
            log(f"d256 [{prec}]: {out['d256'][prec]['ms_per_rollout']:.2f} ms per rollout")
    if world == 1 and not args.no_d512 and args.d_model != 512:
        out["d512"] = {"f16x3": measure_d256(args, device, batch, "f16x3", d=512, F=128, train=False)}
        log(f"d512 [f16x3]: {out['d512']['f16x3']['ms_per_rollout']:.2f} ms per rollout ({out['d512']['f16x3']['roofline']['kernel']})")
    if world == 1 and not args.no_eig:
        try:
            out["eig"] = measure_eig(device)
            log(f"eig step kernels: location {out['eig']['location']['frac']:.2f}, ces {out['eig']['ces']['frac']:.2f} of the HBM peak")
        except Exception as e:      # a secondary leg must not take the headline line with it
            out["eig"] = {"error": repr(e)}
    if dist is not None:          # every collective is behind us: the ranks part here, rank 0 goes on to the CPU baseline
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args, model)
        log("cpu baseline done")
    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())


if __name__ == "__main__":
    main()

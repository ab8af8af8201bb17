"""Rollout throughput on the five BASELINE.json configs, one JSON object per config:
   python tools/config_bench.py [--configs 1,2,3,5] [--steps 3]
cfg1 al_mix dx=1 B=8 T=5 nq=32 | cfg2 location B=1000 T=30 nq=200 (d=32 and d=256/F=1024/H=8 bf16) |
cfg3 al_mix dx=2 B=512 (one GPU's share of 4096) T=50 nq=200, split mask | cfg5 psychometric d=512 H=8 predefined
mask T=30 nq=200 (F=128 literal config and F=2048).  cfg4 (CES EIG) is tools/eig_bench.py.
Random-init weights, synthetic task data, sampled designs, HIP-graph replay; which kernel path ran is reported."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from aline_amd import Aline, Embedder, Encoder, OutputHead  # noqa: E402
from aline_amd.rollout import Rollout  # noqa: E402
from aline_amd.tasks import GPTask, HiddenLocation, PsychometricTask  # noqa: E402
from aline_amd.utils import create_target_mask  # noqa: E402
from bench import PEAK_BF16_DENSE_TFLOPS, PEAK_F32_MFMA_TFLOPS, algorithmic_flops_per_episode  # noqa: E402

# MFMA passes a product costs on the f16/bf16 pipe, by arithmetic policy (f32 runs on the fp32 MFMA instead)
PASSES = {"bf16": 1, "bf16x3": 3, "f16x3": 3}


def build(dx, d, F, H, n_theta, emb, precision):
    m = Aline(Embedder(dx, 1, d, F, n_theta, emb), Encoder(d, F, H, 0.0, 3), OutputHead(dx, 1, d, F)).cuda()
    return m.set_precision(precision).train()


def path_of(d, F, H, emb, precision, keys):
    if precision == "f32" and (d, F, H) == (32, 128, 4) and emb == "theta":
        return "fused::rollout_f32_kernel"
    if precision == "f16x3" and d == 256 and H == 8 and keys <= 64:
        return "x3::layer_kernel"
    if precision == "f16x3" and d == 512 and H == 8 and keys <= 64:
        return "x5::layer_kernel"
    return "generic pipeline"


def roofline(geo, precision, B, dt):
    """Whole-rollout roofline on the FLOPs the kernels execute for the reference's algorithm (SURVEY 8-d formula,
    WITHOUT posterior_out_query: it is lazy and not computed).  `frac` prices the algorithmic FLOPs against the peak of
    the pipe the policy runs on; `frac_executed` counts the 3 passes of a split product (instruction-mix view)."""
    fl = algorithmic_flops_per_episode(with_query_gmm=False, **geo) * B
    peak = PEAK_F32_MFMA_TFLOPS if precision == "f32" else PEAK_BF16_DENSE_TFLOPS
    ach = fl / dt / 1e12
    r = {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": None,
         "algorithmic_flops_per_rollout": fl, "with_query_gmm": False, "scope": "whole rollout graph"}
    if precision in PASSES and PASSES[precision] > 1:
        r["frac_executed"] = ach * PASSES[precision] / peak
    return r


ONLY_PRECS = None


def run(name, model, batch, T, steps, meta):
    if ONLY_PRECS is not None and meta["precision"] not in ONLY_PRECS:
        return
    ro = Rollout(model, batch, T, select="sample", keep_zt=False, keep_posterior=True)
    ro.run()
    torch.cuda.synchronize()
    ro.capture()
    ro.refresh_uniform(); ro.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ro.refresh_uniform()
        ro.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    B, nq = ro.B, ro.n_q0
    out = {"config": name, "ms_per_rollout": dt * 1e3, "designs_per_s": B * T * nq / dt, "B": B, "T": T,
           "n_query_init": nq, "n_tokens": ro.P + ro.n_t, **{k: v for k, v in meta.items() if k != "geo"}, "path": ro.path}
    if "geo" in meta:
        out["roofline"] = roofline(dict(meta["geo"], T=T), meta["precision"], B, dt)
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="1,2,3,5")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--precs", default="", help="comma list: only these precisions (default: all)")
    args = ap.parse_args()
    global ONLY_PRECS
    ONLY_PRECS = set(args.precs.split(",")) if args.precs else None
    want = set(args.configs.split(","))
    dev = torch.device("cuda")
    torch.manual_seed(123)
    if "1" in want:
        task = GPTask(dim_x=1, embedding_type="mix", n_context_init=1, n_query_init=32, n_target_theta=2,
                      n_target_data=100, device=dev)
        geo = dict(dx=1, dy=1, d=32, F=128, H=4, L=3, C=10, n_c0=1, n_q0=32, n_td=100, n_th=2, n_s=102, embedding_type="mix")
        b1 = task.sample_batch(8)
        for prec in ("f32", "f16x3"):
            run(f"cfg1 al_mix dx=1 B=8 T=5 {prec}", build(1, 32, 128, 4, 2, "mix", prec), b1, 5, args.steps,
                {"d": 32, "precision": prec, "path": "", "geo": geo})
    if "2" in want:
        task = HiddenLocation(n_query_init=200, device=dev)
        batch = task.sample_batch(1000)
        geo = dict(dx=2, dy=1, d=32, F=128, H=4, L=3, C=10, n_c0=1, n_q0=200, n_td=0, n_th=2, n_s=2)
        run("cfg2 location_finding B=1000 T=30 d=32", build(2, 32, 128, 4, 2, "theta", "f32"), batch, 30, args.steps,
            {"d": 32, "precision": "f32", "path": path_of(32, 128, 4, "theta", "f32", 32), "geo": geo})
        run("cfg2 location_finding B=1000 T=30 d=32 f16x3", build(2, 32, 128, 4, 2, "theta", "f16x3"), batch, 30, args.steps,
            {"d": 32, "precision": "f16x3", "path": "", "geo": geo})
        for prec in ("f16x3", "bf16"):
            run(f"cfg2 location_finding B=1000 T=30 d=256 F=1024 H=8 {prec}", build(2, 256, 1024, 8, 2, "theta", prec), batch, 30,
                args.steps, {"d": 256, "precision": prec, "path": path_of(256, 1024, 8, "theta", prec, 32),
                             "geo": dict(geo, d=256, F=1024, H=8)})
    if "3" in want:
        task = GPTask(dim_x=2, embedding_type="mix", n_context_init=1, n_query_init=200, n_target_theta=3,
                      n_target_data=100, device=dev)
        batch = task.sample_batch(512)
        batch["target_mask"] = create_target_mask("split", "mix", 100, 3, None, None, None, None, "data")
        geo = dict(dx=2, dy=1, d=32, F=128, H=4, L=3, C=10, n_c0=1, n_q0=200, n_td=100, n_th=3, n_s=100, embedding_type="mix")
        for prec in ("f32", "f16x3"):
            run(f"cfg3 al_mix dx=2 B=512 (of 4096 over 8 GPUs) T=50, split mask (data) {prec}", build(2, 32, 128, 4, 3, "mix", prec),
                batch, 50, args.steps, {"d": 32, "precision": prec, "path": path_of(32, 128, 4, "mix", prec, 999), "geo": geo})
    if "5" in want:
        task = PsychometricTask(n_query_init=200, n_context_init=1, device=dev)
        batch = task.sample_batch(256)
        batch["target_mask"] = torch.tensor([False, False, True, True])
        for F in (128, 2048):
            geo = dict(dx=1, dy=1, d=512, F=F, H=8, L=3, C=10, n_c0=1, n_q0=200, n_td=0, n_th=4, n_s=2)
            for prec in ("f16x3", "bf16"):
                run(f"cfg5 psychometric B=256 T=30 d=512 F={F} H=8, predefined mask {prec}", build(1, 512, F, 8, 4, "theta", prec),
                    batch, 30, args.steps, {"d": 512, "precision": prec, "path": path_of(512, F, 8, "theta", prec, 0), "geo": geo})


if __name__ == "__main__":
    main()

"""Round-4 fixtures, produced by the REFERENCE itself (imported from /root/reference as in make_golden.py; run in the build
container only, the fixtures travel).  They pin the BACKWARD (train_aline.py:113-132: loss = design_loss + predict_loss,
`loss.backward()` of the reference's autograd) where rounds 1-3 had forward fixtures only:

  grad_cfg2_d256   location finding, d = 256 / F = 1024 / 8 heads of 32 (the roofline variant), B = 2, T = 3
  grad_cfg5_d512   psychometric, d = 512 / F = 128 / 8 heads of 64, predefined mask [F, F, T, T], B = 2, T = 3
  grad_cfg3_split  al_mix dim_x = 2, 100 data + 3 theta targets, split mask on the data targets, B = 2, T = 4
  grad_cfg4_ces    CES dim_x = 6, 5 theta targets, B = 3, T = 4
  eval_boed_loc    utils/eval.py:142-198 (`eval_boed`, stepwise) on location finding with every random draw of the call
                   recorded: the initial batch, the prior contrastive samples, and the reference's own designs.

A wide model has 6 - 23 M parameters, so a gradient fixture does not store whole tensors: per parameter it keeps the L2 norm,
the sum, the max |g| and the values at <= 4096 seeded positions (`gidx.<name>` / `gval.<name>`); tensors of <= 8192 elements are
kept whole (`grad.<name>`).
    python oracle/make_golden_r4.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import make_golden as mg          # noqa: E402  (sets up the reference imports)

WHOLE, NSAMP = 8192, 4096


def compress_grads(arrays, prefix="train"):
    """<prefix>.grad.<k> -> whole tensor (small) or norm / sum / max + seeded sample (large; the same positions for every prefix)."""
    out = {}
    gen = torch.Generator().manual_seed(20260405)
    for k in sorted(arrays):
        v = arrays[k]
        if not k.startswith(prefix + ".grad."):
            out[k] = v
            continue
        name = k[len(prefix + ".grad."):]
        g = torch.from_numpy(v).reshape(-1).double()
        out[f"{prefix}.gnorm." + name] = np.float64(g.norm().item())
        out[f"{prefix}.gsum." + name] = np.float64(g.sum().item())
        out[f"{prefix}.gmax." + name] = np.float64(g.abs().max().item())
        if g.numel() <= WHOLE:
            out[k] = v
        else:
            idx = torch.randperm(g.numel(), generator=gen)[:NSAMP].sort().values
            out[f"{prefix}.gidx." + name] = idx.numpy().astype(np.int64)
            out[f"{prefix}.gval." + name] = v.reshape(-1)[idx.numpy()]
    return out


class _Forced:
    """Teacher forcing for the reference (it has none): model/head.py:351-353 draws the design from `Categorical(zt).sample()`;
    while active, the `Categorical` name of the reference's head module is a subclass whose `sample()` returns the recorded
    designs step by step.  Nothing else of the reference is touched."""

    def __init__(self, forced):
        self.forced, self.t = forced, 0

    def __enter__(self):
        import model.head as mh
        outer = self

        class Cat(torch.distributions.Categorical):
            def sample(self, *a, **k):
                idx = outer.forced[:, outer.t].clone()
                outer.t += 1
                return idx
        self.mh, self.old = mh, mh.Categorical
        mh.Categorical = Cat
        return self

    def __exit__(self, *exc):
        self.mh.Categorical = self.old


def gen_grad_fixture(name, task, dims, B, T, mask_kwargs, mask_type="all", seed=123, wseed=7):
    mg.seed_all(seed)
    model = mg.build_model(dims, wseed)
    batch = task.sample_batch(B)
    batch.target_mask = mg.ref_mask.create_target_mask(**mask_kwargs)
    arrays = {}
    mg.batch_to_np(batch, arrays)
    mg.seed_all(seed + 1)
    tr = mg.run_rollout(model, task, batch, T, "train", dims["embedding_type"], mask_type, dims["n_theta"],
                        with_grads=True, time_token=dims.get("time_token", False))
    keep = ("idx_", "target_ll_", "log_probs", "nll", "R", "design_loss", "predict_loss", "grad.", "final_context")
    for k, v in tr.items():
        if k.startswith(keep):
            arrays["train." + k] = v
    arrays = compress_grads(arrays)
    # the same rollout by the reference in fp64 (teacher-forced with the fp32 run's designs): what the fp32 gradients above are
    # an approximation of -- at CES and at d = 256 the reference's own fp32 rounding is ~1e-3 of a parameter's max |grad|
    forced = torch.from_numpy(np.concatenate([tr[f"idx_{t}"] for t in range(T)], axis=1))
    model64 = mg.build_model(dims, wseed).double()
    b64 = mg.AttrDict({k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in batch.items()})
    torch.set_default_dtype(torch.float64)
    try:
        with _Forced(forced):
            tr64 = mg.run_rollout(model64, task, b64, T, "train", dims["embedding_type"], mask_type, dims["n_theta"],
                                  with_grads=True, time_token=dims.get("time_token", False))
    finally:
        torch.set_default_dtype(torch.float32)
    a64 = {"train64." + k: v for k, v in tr64.items() if k.startswith(("grad.", "R", "design_loss", "predict_loss", "nll"))}
    for t in range(T):
        a64[f"train64.target_ll_{t}"] = tr64[f"target_ll_{t}"]
    arrays.update(compress_grads(a64, prefix="train64"))
    meta = dict(dims=dims, B=B, T=T, seed=seed, wseed=wseed, mask_type=mask_type, n_c0=int(batch.context_x.shape[1]),
                n_q0=int(batch.query_x.shape[1]), n_t=int(batch.target_all.shape[1]))
    mg.save(name, meta, arrays)


def gen_eval_boed_fixture():
    """`eval_boed` (utils/eval.py:142-198, stepwise) with every random draw of the call recorded in call order: the reference
    task's `sample_batch` (eval.py:21) and `sample_theta` (eval.py:18 shape probe, eval.py:61 contrastive draw) are wrapped by
    recorders, `get_traces` by one that keeps the histories.  The product replays the tape through the same two methods."""
    from tasks.location_finding import HiddenLocation
    mg.seed_all(77)
    task = HiddenLocation(n_query_init=40)
    dims = mg.dims_of(2, n_theta=2)
    model = mg.build_model(dims, 7)
    model.eval()
    B, T, L, M = 6, 5, 64, 12
    arrays, n = {}, {"batch": 0, "theta": 0, "trace": 0}
    orig_batch, orig_theta, orig_traces = task.sample_batch, task.sample_theta, mg.ref_eval.get_traces

    def rec_batch(batch_size):
        b = orig_batch(batch_size)
        mg.batch_to_np(b, arrays, prefix=f"batch{n['batch']}.")
        n["batch"] += 1
        return b

    def rec_theta(shape):
        th = orig_theta(shape)
        arrays[f"theta{n['theta']}"] = th.numpy().copy()
        n["theta"] += 1
        return th

    def rec_traces(*a, **k):
        th0, x, y = orig_traces(*a, **k)
        i = n["trace"]
        arrays[f"trace{i}.theta0"], arrays[f"trace{i}.x"], arrays[f"trace{i}.y"] = th0.numpy().copy(), x.numpy().copy(), y.numpy().copy()
        n["trace"] += 1
        return th0, x, y

    task.sample_batch, task.sample_theta, mg.ref_eval.get_traces = rec_batch, rec_theta, rec_traces
    mg.seed_all(78)
    bounds = mg.ref_eval.eval_boed(model, task, T=T, L=L, M=M, batch_size=B, time_token=False, stepwise=True, err_type="se")
    task.sample_batch, task.sample_theta, mg.ref_eval.get_traces = orig_batch, orig_theta, orig_traces
    for k in ("pce_mean", "pce_err", "nmc_mean", "nmc_err"):
        arrays[k] = bounds[k].numpy()
    mg.save("eval_boed_loc", dict(dims=dims, wseed=7, B=B, T=T, L=L, M=M, n_q0=40, n_batch=n["batch"], n_theta=n["theta"]), arrays)


def main():
    torch.set_num_threads(8)
    from tasks.location_finding import HiddenLocation
    from tasks.gaussian_process import GPTask
    from tasks.ces import CESTask
    from tasks.psychometric import PsychometricTask
    d2b = mg.dims_of(2, d=256, F=1024, n_head=8, n_theta=2)
    gen_grad_fixture("grad_cfg2_d256", HiddenLocation(), d2b, B=2, T=3, mask_kwargs=mg.mask_kw("all", "theta", 0, 2))
    d5 = mg.dims_of(1, d=512, F=128, n_head=8, n_theta=4)
    gen_grad_fixture("grad_cfg5_d512", PsychometricTask(n_context_init=1, n_query_init=200), d5, B=2, T=3,
                     mask_type="predefined",
                     mask_kwargs=mg.mask_kw("predefined", "theta", 0, 4, predefined=[[False, False, True, True],
                                                                                    [True, True, False, False]], mask_index=0))
    d3 = mg.dims_of(2, n_theta=3, embedding_type="mix")
    gp3 = dict(dim_x=2, embedding_type="mix", n_context_init=1, n_query_init=200, n_target_theta=3, n_target_data=100,
               design_scale=5, noise_scale=0.01)
    gen_grad_fixture("grad_cfg3_split", GPTask(**gp3), d3, B=2, T=4, mask_type="split",
                     mask_kwargs=mg.mask_kw("split", "mix", 100, 3, attend_to="data"))
    d4 = mg.dims_of(6, n_theta=5)
    gen_grad_fixture("grad_cfg4_ces", CESTask(n_context_init=1, n_query_init=200), d4, B=3, T=4,
                     mask_kwargs=mg.mask_kw("all", "theta", 0, 5))
    gen_eval_boed_fixture()


if __name__ == "__main__":
    main()

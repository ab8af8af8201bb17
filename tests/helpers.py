"""Shared by the GPU parity tests: build the native model from the deterministic weights and run it."""
import torch

import aline_oracle as orc


def native_model(dims, wseed, precision="f32", device="cuda"):
    from aline_amd import Aline, Embedder, Encoder, OutputHead
    model = Aline(
        Embedder(dims["dim_x"], dims["dim_y"], dims["d"], dims["F"], dims["n_theta"],
                 dims["embedding_type"]),
        Encoder(dims["d"], dims["F"], dims["n_head"], 0.0, dims["L"]),
        OutputHead(dims["dim_x"], dims["dim_y"], dims["d"], dims["F"], num_components=dims["C"],
                   time_token=dims.get("time_token", False)))
    sd = orc.make_state_dict(wseed, **dims)
    model.load_state_dict(sd, strict=True)          # state_dict key compatibility
    return model.to(device).set_precision(precision), sd


def to_dev(batch, device="cuda"):
    from aline_amd.utils import AttrDict
    return AttrDict({k: (v.to(device) if torch.is_tensor(v) else v) for k, v in batch.items()})


def maxdiff(a, b):
    return float((a.detach().cpu().float() - torch.as_tensor(b).float()).abs().max())


def grad_errors(fx, named_grads, prefix="train"):
    """Per-parameter error of gradients against a fixture's reference-autograd gradients, relative to the parameter's max |grad|
    (floor 1e-4).  Round-1/3 fixtures keep whole tensors (`train.grad.<k>`); the wide round-4 ones (oracle/make_golden_r4.py)
    keep the L2 norm, the sum, max |g| and <= 4096 seeded positions per large tensor, once from the reference's fp32 run (`train`)
    and once from the reference run in fp64 on the same designs (`train64`).  Returns {name: relative error}."""
    out = {}
    for k, got in named_grads:
        got = got.detach().cpu().double().reshape(-1)
        if f"{prefix}.grad." + k in fx:
            ref = fx.t(f"{prefix}.grad." + k).double().reshape(-1)
            scale = max(float(ref.abs().max()), 1e-4)
            out[k] = float((got - ref).abs().max()) / scale
        else:
            idx, val = fx.t(f"{prefix}.gidx." + k), fx.t(f"{prefix}.gval." + k).double()
            gmax, nref = float(fx.np(f"{prefix}.gmax." + k)), float(fx.np(f"{prefix}.gnorm." + k))
            scale = max(gmax, 1e-4)
            err = float((got[idx] - val).abs().max()) / scale
            # whole-tensor statistics: L2 norm (relative) and max |g|
            err = max(err, abs(float(got.norm()) - nref) / max(nref, 1e-4 * got.numel() ** 0.5))
            err = max(err, abs(float(got.abs().max()) - gmax) / scale)
            out[k] = err
    return out

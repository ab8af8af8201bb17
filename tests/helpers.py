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

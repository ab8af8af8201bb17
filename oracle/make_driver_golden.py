#!/usr/bin/env python3
"""Golden data for the training-driver helpers: runs the REFERENCE's own utils/misc.py
(`set_layerwise_lr`, `save_checkpoint`) on the reference `Aline` model and records what callers and checkpoint files
depend on: optimiser class, parameter names and learning rate per group, scheduler horizon, the lr trajectory over a few
steps, the checkpoint file name and its keys.  Build container only (the reference never travels); writes
tests/golden/driver.json.  utils/misc.py imports hydra / omegaconf at module level without using them in these
functions; like `attrdictionary` they are absent third-party packages and are stubbed.

    python oracle/make_driver_golden.py
"""
import importlib.util
import json
import os
import sys
import tempfile
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden", "driver.json")
sys.dont_write_bytecode = True


class AttrDict(dict):
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.__dict__ = self


for name, attrs in (("attrdictionary", {"AttrDict": AttrDict}), ("omegaconf", {"OmegaConf": object}),
                    ("hydra", {"initialize_config_dir": None, "compose": None})):
    mod = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(mod, k, v)
    sys.modules[name] = mod
sys.path.insert(0, "/root/reference")
from model.base import Aline  # noqa: E402
from model.embedder import Embedder  # noqa: E402
from model.encoder import Encoder  # noqa: E402
from model.head import OutputHead  # noqa: E402

spec = importlib.util.spec_from_file_location("ref_misc", "/root/reference/utils/misc.py")
misc = importlib.util.module_from_spec(spec)
spec.loader.exec_module(misc)

torch.manual_seed(0)
model = Aline(Embedder(2, 1, 32, 128, 2, "theta"), Encoder(32, 128, 4, 0.0, 3), OutputHead(2, 1, 32, 128))
names = {id(p): n for n, p in model.named_parameters()}
out = {"param_names": [n for n, _ in model.named_parameters()], "cases": []}
for optimizer in ("AdamW", "Adam"):
    for epoch in (0, 3, 4, 9):
        cfg = AttrDict(optimizer=optimizer, lr=1e-3, max_epoch=12, burning_epoch=4)
        opt, sch = misc.set_layerwise_lr(cfg, model, epoch)
        groups = [{"lr": g["lr"], "names": [names[id(p)] for p in g["params"]]} for g in opt.param_groups]
        traj = []
        for _ in range(5):
            opt.step()
            sch.step()
            traj.append([g["lr"] for g in opt.param_groups])
        out["cases"].append({"optimizer": optimizer, "epoch": epoch, "class": type(opt).__name__, "groups": groups,
                             "T_max": sch.T_max, "lr_after_steps": traj})
with tempfile.TemporaryDirectory() as tmp:
    cfg = AttrDict(optimizer="AdamW", lr=1e-3, max_epoch=12, burning_epoch=4, output_dir=tmp, checkpoint_name="ckpt.tar")
    opt, sch = misc.set_layerwise_lr(cfg, model, 5)
    misc.save_checkpoint(cfg, model, opt, sch, 6, with_epoch=True)
    misc.save_checkpoint(cfg, model, opt, sch, 6, with_epoch=False)
    files = sorted(os.listdir(tmp))
    state = torch.load(os.path.join(tmp, "ckpt_6.tar"), weights_only=False)
    out["checkpoint"] = {"files": files, "keys": sorted(state.keys()), "epoch": state["epoch"],
                         "optimizer_group_sizes": [len(g["params"]) for g in state["optimizer"]["param_groups"]],
                         "model_keys": sorted(state["model"].keys())}
    path = misc.save_state_dict(model, tmp, "aae_x_burning.pth")
    out["state_dict_path"] = os.path.relpath(path, tmp)
json.dump(out, open(OUT, "w"), indent=1)
print("wrote", OUT, len(out["cases"]), "cases")

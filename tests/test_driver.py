"""Training driver: optimiser / scheduler construction, checkpoint format and the epoch loop
across the burn-in boundary.  CPU tests cover the host logic; the loop itself runs on the GPU."""
import os
import random

import numpy as np
import pytest
import torch
from torch import nn


class _Cfg(dict):
    __getattr__ = dict.get


def _cfg(tmp, **kw):
    c = _Cfg(optimizer="AdamW", lr=1e-3, max_epoch=10, burning_epoch=4, checkpoint=0, checkpoint_name="ckpt.tar",
             output_dir=str(tmp), file_name="aae_test.pth", T=5, min_T=5, alpha=1.0, gamma=1.0, clip_grads=True)
    c.update(kw)
    return c


class _Toy(nn.Module):
    """Parameter names as in the reference: the acquisition MLP lives under `...predictor...`."""

    def __init__(self):
        super().__init__()
        self.encoder = nn.Linear(4, 4)
        self.head = nn.ModuleDict({"acquisition_head": nn.ModuleDict({"predictor": nn.Linear(4, 1)}),
                                   "target_head": nn.Linear(4, 3)})


def test_layerwise_lr_before_and_after_burn_in(tmp_path):
    from aline_amd.driver import set_layerwise_lr
    model, cfg = _Toy(), _cfg(tmp_path)
    opt, sch = set_layerwise_lr(cfg, model, epoch=0)               # misc.py:145-152
    assert type(opt).__name__ == "AdamW" and len(opt.param_groups) == 1
    assert opt.param_groups[0]["lr"] == 1e-3 and sch.T_max == 10
    opt, sch = set_layerwise_lr(cfg, model, epoch=4)               # misc.py:153-170
    shared, pred = opt.param_groups
    assert shared["lr"] == pytest.approx(2e-4) and pred["lr"] == 1e-3 and sch.T_max == 6
    pred_ids = {id(p) for n, p in model.named_parameters() if "predictor" in n}
    assert {id(p) for p in pred["params"]} == pred_ids and len(pred_ids) == 2
    assert len(shared["params"]) == len(list(model.parameters())) - 2
    # other optimisers by name, as getattr(optim, cfg.optimizer)
    opt, _ = set_layerwise_lr(_cfg(tmp_path, optimizer="SGD"), model, epoch=0)
    assert type(opt).__name__ == "SGD"


def test_checkpoint_round_trip_and_format(tmp_path):
    from aline_amd.driver import load_checkpoint, save_checkpoint, set_layerwise_lr
    torch.manual_seed(0)
    model, cfg = _Toy(), _cfg(tmp_path, burning_epoch=2)
    opt, sch = set_layerwise_lr(cfg, model, epoch=3)               # past burn-in: two parameter groups
    for _ in range(3):
        opt.zero_grad()
        sum(p.square().sum() for p in model.parameters()).backward()
        opt.step()
        sch.step()
    random.seed(5); np.random.seed(6); torch.manual_seed(7)
    path = save_checkpoint(cfg, model, opt, sch, epoch=4, with_epoch=True)
    assert os.path.basename(path) == "ckpt_4.tar"                  # misc.py:84-87
    state = torch.load(path, weights_only=False)
    assert set(state) == {"model", "optimizer", "scheduler", "epoch", "rng_state", "cuda_rng_state",
                          "numpy_rng_state", "random_rng_state"}   # misc.py:72-82
    expect = (random.random(), float(np.random.rand()), float(torch.rand(1)))
    want = {k: v.clone() for k, v in model.state_dict().items()}
    lr_saved = [g["lr"] for g in opt.param_groups]

    other = _Toy()
    o2, s2 = set_layerwise_lr(cfg, other, epoch=0)                 # wrong (burn-in) groups on purpose
    epoch, o2, s2 = load_checkpoint(cfg, other, o2, s2, ckpt_path=path)
    assert epoch == 4 and len(o2.param_groups) == 2
    assert [g["lr"] for g in o2.param_groups] == pytest.approx(lr_saved)
    assert s2.last_epoch == sch.last_epoch
    for k, v in other.state_dict().items():
        assert torch.equal(v, want[k])
    assert (random.random(), float(np.random.rand()), float(torch.rand(1))) == expect   # RNG streams restored
    with pytest.raises(FileNotFoundError):
        load_checkpoint(cfg, other, o2, s2, ckpt_path=str(tmp_path / "missing.tar"))


def test_state_dict_files(tmp_path):
    from aline_amd.driver import load_state_dict, save_state_dict
    a, b = _Toy(), _Toy()
    path = save_state_dict(a, str(tmp_path), "aae_test_burning.pth")
    assert path == os.path.join(str(tmp_path), "model", "aae_test_burning.pth")     # misc.py:39-43
    load_state_dict(b, str(tmp_path), "aae_test_burning.pth")
    for (k, v), (_, w) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(v, w), k


def test_epoch_target_mask_draws_from_the_configured_types():
    from aline_amd.driver import epoch_target_mask
    random.seed(0)
    task = _Cfg(mask_type=["all"], embedding_type="theta", n_target_data=0, n_target_theta=3)
    assert epoch_target_mask(task)[1].tolist() == [True] * 3
    task = _Cfg(mask_type=["split"], embedding_type="mix", n_target_data=4, n_target_theta=2, attend_to="theta")
    mt, mask = epoch_target_mask(task)
    assert mt == "split" and mask.tolist() == [False] * 4 + [True] * 2
    task = _Cfg(mask_type=["predefined"], embedding_type="theta", n_target_data=0, n_target_theta=4,
                predefined_masks=[[False, False, True, True], [True, True, False, False]],
                predefined_mask_weights=[1, 1])
    seen = {tuple(epoch_target_mask(task)[1].tolist()) for _ in range(40)}
    assert seen == {(False, False, True, True), (True, True, False, False)}


@pytest.mark.gpu
def test_train_loop_across_burn_in_and_resume(tmp_path):
    """6 epochs with burn-in 3 on a small location_finding model: candidate-set size and parameter groups switch
    at the boundary, the burn-in state dict and the epoch checkpoints are written, and a resumed run continues
    from the checkpoint with the post-burn-in optimiser."""
    from aline_amd import Aline, Embedder, Encoder, OutputHead
    from aline_amd.driver import train
    from aline_amd.tasks import HiddenLocation
    torch.manual_seed(0); random.seed(0)
    dev = torch.device("cuda")
    model = Aline(Embedder(2, 1, 32, 128, 2, "theta"), Encoder(32, 128, 4, 0.0, 3), OutputHead(2, 1, 32, 128)).to(dev)
    task = HiddenLocation(n_query_init=40, device=dev)
    cfg = _cfg(tmp_path, max_epoch=6, burning_epoch=3, checkpoint=2, T=6, min_T=4, batch_size=16,
               task=_Cfg(mask_type=["all"], embedding_type="theta", n_target_data=0, n_target_theta=2,
                         n_query_init=40))
    seen_nq, acq_drift = [], []
    acq0 = [p.detach().clone() for p in model.head.acquisition_head.parameters()]

    def on_epoch(r):
        seen_nq.append(task.n_query_init)
        acq_drift.append(max(float((p.detach() - q).abs().max()) for p, q in zip(model.head.acquisition_head.parameters(), acq0)))
    recs = train(cfg, model, task, on_epoch=on_epoch)
    assert [r["epoch"] for r in recs] == list(range(6))
    assert all(4 <= r["T"] <= 6 for r in recs) and all(np.isfinite(r["loss"]) for r in recs)
    # burn-in: n_query_init = T, prediction loss only; afterwards the configured candidate set and two lr groups
    assert seen_nq == [6, 6, 6, 40, 40, 40]
    # during burn-in no loss term reaches the acquisition head (train_aline.py:126-128): its .grad is None in the reference and
    # AdamW leaves it alone -- no weight decay, no moment updates; from the first design-loss epoch on it moves
    assert acq_drift[:3] == [0.0, 0.0, 0.0] and acq_drift[3] > 0.0, acq_drift
    assert all(r["loss"] == pytest.approx(r["predict_loss"]) for r in recs[:3])
    assert [len(r["lr"]) for r in recs] == [1, 1, 1, 2, 2, 2]
    assert recs[3]["lr"][0] == pytest.approx(recs[3]["lr"][1] / 5)
    assert os.path.exists(tmp_path / "model" / "aae_test_burning.pth")
    assert all(os.path.exists(tmp_path / f"ckpt_{e}.tar") for e in (2, 4, 6))
    # resume from epoch 4
    model2 = Aline(Embedder(2, 1, 32, 128, 2, "theta"), Encoder(32, 128, 4, 0.0, 3), OutputHead(2, 1, 32, 128)).to(dev)
    cfg2 = _cfg(tmp_path, **{**cfg, "load_checkpoint": True, "load_path": str(tmp_path / "ckpt_4.tar"), "checkpoint": 0})
    recs2 = train(cfg2, model2, task)
    assert [r["epoch"] for r in recs2] == [4, 5] and [len(r["lr"]) for r in recs2] == [2, 2]
    assert recs2[0]["lr"] == pytest.approx(recs[4]["lr"])


def test_driver_helpers_match_the_reference(tmp_path):
    """tests/golden/driver.json holds what the reference's own utils/misc.py produced for the reference Aline model
    (oracle/make_driver_golden.py): optimiser class, parameter names + lr per group, scheduler horizon, lr trajectory,
    checkpoint file names / keys.  The drop-in model has the same parameter names, so everything must coincide."""
    import json
    from aline_amd import Aline, Embedder, Encoder, OutputHead
    from aline_amd.driver import save_checkpoint, save_state_dict, set_layerwise_lr
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "driver.json")))
    model = Aline(Embedder(2, 1, 32, 128, 2, "theta"), Encoder(32, 128, 4, 0.0, 3), OutputHead(2, 1, 32, 128))
    assert [n for n, _ in model.named_parameters()] == gold["param_names"]
    names = {id(p): n for n, p in model.named_parameters()}
    for case in gold["cases"]:
        cfg = _cfg(tmp_path, optimizer=case["optimizer"], lr=1e-3, max_epoch=12, burning_epoch=4)
        opt, sch = set_layerwise_lr(cfg, model, case["epoch"])
        assert type(opt).__name__ == case["class"] and sch.T_max == case["T_max"]
        assert len(opt.param_groups) == len(case["groups"])
        for g, ref in zip(opt.param_groups, case["groups"]):
            assert g["lr"] == pytest.approx(ref["lr"]) and [names[id(p)] for p in g["params"]] == ref["names"]
        for ref_lrs in case["lr_after_steps"]:
            opt.step()
            sch.step()
            assert [g["lr"] for g in opt.param_groups] == pytest.approx(ref_lrs)
    cfg = _cfg(tmp_path, max_epoch=12, burning_epoch=4)
    opt, sch = set_layerwise_lr(cfg, model, 5)
    save_checkpoint(cfg, model, opt, sch, 6, with_epoch=True)
    save_checkpoint(cfg, model, opt, sch, 6, with_epoch=False)
    ck = gold["checkpoint"]
    assert sorted(f for f in os.listdir(tmp_path) if f.endswith(".tar")) == ck["files"]
    state = torch.load(tmp_path / "ckpt_6.tar", weights_only=False)
    assert sorted(state.keys()) == ck["keys"] and state["epoch"] == ck["epoch"]
    assert [len(g["params"]) for g in state["optimizer"]["param_groups"]] == ck["optimizer_group_sizes"]
    assert sorted(state["model"].keys()) == ck["model_keys"]
    path = save_state_dict(model, str(tmp_path), "aae_x_burning.pth")
    assert os.path.relpath(path, str(tmp_path)) == gold["state_dict_path"]

"""Training driver around the native training step: the epoch loop of the reference's
`train()` (train_aline.py:21-181) without its hydra / wandb shell, plus the optimiser / checkpoint helpers
it relies on (utils/misc.py:30-171).  Behaviour that callers and existing checkpoints depend on is kept:

* burn-in: the first `cfg.burning_epoch` epochs optimise the prediction loss only and roll out with
  `n_query_init = cfg.T` candidate designs (train_aline.py:47-48, 126-132); at `epoch == burning_epoch` the
  optimiser is rebuilt AFTER the backward pass and BEFORE the step (train_aline.py:142-149);
* optimiser: `getattr(torch.optim, cfg.optimizer)` at `cfg.lr` with cosine annealing over `max_epoch` during
  burn-in; afterwards two parameter groups -- every parameter whose name does not contain 'predictor' at
  `lr / 5`, the acquisition predictor at `lr` -- annealed over `max_epoch - burning_epoch` (misc.py:137-171);
* checkpoints: `<checkpoint_name stem>_<epoch>.tar` holding model / optimizer / scheduler state, the epoch
  and the torch / cuda / numpy / python RNG states (misc.py:61-89); loading rebuilds the optimiser for
  `epoch - 1` first so that the saved parameter groups fit (misc.py:92-135).  Files written by the reference
  load here and vice versa (same keys, same state_dict names).

The per-epoch work itself is `aline_amd.train.train_step`: fused rollout, REINFORCE terms, native backward,
one flat-bucket all-reduce under episode data parallelism, inf-norm clipping.
"""
import os
import random
import time

import numpy as np
import torch
from torch import optim
from torch.optim import lr_scheduler

from .train import check_range_async, check_training_range, optimizer_step, train_step
from .utils.target_mask import create_target_mask


def _get(cfg, name, default=None):
    if isinstance(cfg, dict):
        return cfg.get(name, default)
    return getattr(cfg, name, default)


def set_layerwise_lr(cfg, model, epoch=0):
    """Optimiser + cosine scheduler for `epoch` (before / after the burn-in boundary)."""
    opt_cls = getattr(optim, _get(cfg, "optimizer", "AdamW"))
    lr, max_epoch, burn = _get(cfg, "lr"), _get(cfg, "max_epoch"), _get(cfg, "burning_epoch", 0)
    if epoch < burn:
        optimizer = opt_cls(model.parameters(), lr=lr)
        horizon = max_epoch
    else:
        shared = [p for n, p in model.named_parameters() if "predictor" not in n]
        predictor = [p for n, p in model.named_parameters() if "predictor" in n]
        optimizer = opt_cls([{"params": shared, "lr": lr / 5}, {"params": predictor}], lr=lr)
        horizon = max_epoch - burn
    return optimizer, lr_scheduler.CosineAnnealingLR(optimizer, T_max=horizon)


def save_state_dict(model, out_dir, name="aline.pth"):
    """`<out_dir>/model/<name>` = model.state_dict() (misc.py:30-44)."""
    folder = os.path.join(out_dir, "model")
    os.makedirs(folder, exist_ok=True)
    path = os.path.join(folder, name)
    torch.save(model.state_dict(), path)
    return path


def load_state_dict(model, out_dir, name="aline.pth", map_location=None):
    path = os.path.join(out_dir, "model", name)
    model.load_state_dict(torch.load(path, map_location=map_location, weights_only=True))
    return model


def checkpoint_path(cfg, epoch=None):
    name = _get(cfg, "checkpoint_name", "ckpt.tar")
    if epoch is not None:
        name = f"{name.split('.')[0]}_{epoch}.tar"
    return os.path.join(_get(cfg, "output_dir", "."), name)


def save_checkpoint(cfg, model, optimizer, scheduler, epoch, with_epoch=False):
    state = {
        "model": model.state_dict(),
        "optimizer": optimizer.state_dict(),
        "scheduler": scheduler.state_dict(),
        "epoch": epoch,
        "rng_state": torch.get_rng_state(),
        "cuda_rng_state": torch.cuda.get_rng_state() if torch.cuda.is_available() else None,
        "numpy_rng_state": np.random.get_state(),
        "random_rng_state": random.getstate(),
    }
    path = checkpoint_path(cfg, epoch if with_epoch else None)
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    torch.save(state, path)
    return path


def _as_byte_tensor(t):
    return t if isinstance(t, torch.ByteTensor) else torch.ByteTensor(t.cpu())


def load_checkpoint(cfg, model, optimizer, scheduler, ckpt_path=None, check_layerwise=True, map_location=None):
    """Returns (epoch, optimizer, scheduler); restores the four RNG streams."""
    path = ckpt_path or checkpoint_path(cfg)
    if not os.path.exists(path):
        raise FileNotFoundError(f"Checkpoint file not found: {path}")
    state = torch.load(path, map_location=map_location, weights_only=False)
    model.load_state_dict(state["model"])
    epoch = state["epoch"]
    if check_layerwise:     # parameter groups as they were when the checkpoint was written (its last epoch)
        optimizer, scheduler = set_layerwise_lr(cfg, model, epoch - 1)
    optimizer.load_state_dict(state["optimizer"])
    scheduler.load_state_dict(state["scheduler"])
    torch.set_rng_state(_as_byte_tensor(state["rng_state"]))
    if torch.cuda.is_available() and state.get("cuda_rng_state") is not None:
        torch.cuda.set_rng_state(_as_byte_tensor(state["cuda_rng_state"]))
    np.random.set_state(state["numpy_rng_state"])
    random.setstate(state["random_rng_state"])
    return epoch, optimizer, scheduler


def epoch_target_mask(task_cfg):
    """Mask type drawn per epoch from cfg.task.mask_type, then the mask itself (train_aline.py:62-72)."""
    mask_type = random.choice(list(_get(task_cfg, "mask_type", ["all"])))
    mask = create_target_mask(mask_type, _get(task_cfg, "embedding_type", "theta"),
                              _get(task_cfg, "n_target_data", 0), _get(task_cfg, "n_target_theta", 0),
                              _get(task_cfg, "n_selected_targets", None), _get(task_cfg, "predefined_masks", None),
                              _get(task_cfg, "predefined_mask_weights", None), _get(task_cfg, "mask_index", None),
                              _get(task_cfg, "attend_to", None))
    return mask_type, mask


class RankRng:
    """Episode data parallelism: every rank draws its OWN episodes but the SAME per-epoch T and target mask.

    T and the mask type come from python's `random` (train_aline.py:59,62), which stays identical on all ranks.  The mask
    itself may consume torch's generator (`partial`: torch.randperm, weighted `predefined`: torch.multinomial,
    utils/target_mask.py:18,24), and the episodes consume torch / numpy: so the torch stream that was common to all ranks at
    start-up (same seed, or the state load_checkpoint restored) is kept aside as the SHARED stream for mask draws, and the
    global torch / numpy generators are re-seeded per rank for the episodes -- with the start epoch folded in, so that a
    resumed multi-rank run does not replay the episode stream of epochs 0..N of the original run.  world == 1: nothing is
    touched (the single-process run consumes the generators exactly like the reference)."""

    def __init__(self, rank, world, start_epoch=0):
        self.shared = None
        if world <= 1:
            return
        self.shared = torch.get_rng_state()
        base = int(torch.initial_seed()) % (2 ** 31)
        torch.manual_seed(base + 1000003 * (rank + 1) + 7907 * start_epoch)              # CPU and CUDA generators
        np.random.seed((base + 7919 * (rank + 1) + 104729 * start_epoch) % (2 ** 32))

    def shared_draw(self, fn):
        """fn() with torch's global CPU generator switched to the stream all ranks share."""
        if self.shared is None:
            return fn()
        mine = torch.get_rng_state()
        torch.set_rng_state(self.shared)
        try:
            return fn()
        finally:
            self.shared = torch.get_rng_state()
            torch.set_rng_state(mine)


def offset_rank_rng(rank, world, start_epoch=0):
    """(kept for callers of round 2) re-seeds the per-rank generators; see RankRng."""
    return RankRng(rank, world, start_epoch)


def train(cfg, model, experiment, batch_size=None, min_T=None, max_T=None, max_epoch=None, verbose=None,
          logger=None, dist=None, world=1, on_epoch=None, rank=0):
    """Epoch loop of train_aline.py:21-181 on the native training step.  Returns the per-epoch records
    (dicts with epoch, T, mask_type, loss, design_loss, predict_loss, lr, seconds).  With `dist` / `world` > 1 (one
    process per GPU) every rank trains on its own episodes, gradients are all-reduced once per step; checkpoints and
    state_dicts are written by rank 0 only, followed by a barrier."""
    batch_size = batch_size or _get(cfg, "batch_size")
    max_T = max_T or _get(cfg, "T")
    min_T = min_T or _get(cfg, "min_T", max_T)
    max_epoch = max_epoch or _get(cfg, "max_epoch")
    verbose = verbose or _get(cfg, "verbose", 500)
    burn = _get(cfg, "burning_epoch", 0)
    task_cfg = _get(cfg, "task")
    say = logger.info if logger is not None else (lambda *_: None)

    optimizer, scheduler = set_layerwise_lr(cfg, model)
    start_epoch = 0
    if _get(cfg, "load_checkpoint", False):
        start_epoch, optimizer, scheduler = load_checkpoint(cfg, model, optimizer, scheduler, _get(cfg, "load_path"))
    rng = RankRng(rank, world, start_epoch)

    def barrier():
        if dist is not None and world > 1:
            dist.barrier()
    full_n_query = _get(task_cfg, "n_query_init", getattr(experiment, "n_query_init", None))
    if start_epoch < burn:
        experiment.n_query_init = _get(cfg, "T")          # fewer candidates while only the predictor trains

    records = []
    for epoch in range(start_epoch, max_epoch):
        tic = time.time()
        T = random.randint(min_T, max_T)
        batch = experiment.sample_batch(batch_size)
        mask_type, batch["target_mask"] = rng.shared_draw(lambda: epoch_target_mask(task_cfg))
        terms, _ = train_step(model, batch, T, optimizer=None, embedding_type=_get(task_cfg, "embedding_type", "theta"),
                              mask_type=mask_type, gamma=_get(cfg, "gamma", 1.0), alpha=_get(cfg, "alpha", 1.0),
                              burn_in=epoch < burn, clip_grads=_get(cfg, "clip_grads", True), dist=dist, world=world)
        # f16x3 range guard: stop BEFORE the update is applied or a checkpoint written, on all ranks together (the status went
        # through the step's gradient all-reduce).  The loop synchronises at float(terms["loss"]) below anyway.
        check_training_range(terms)
        if epoch == burn:
            # from here on the shared layers learn at lr / 5; this epoch's gradients are applied by the new optimiser
            optimizer, scheduler = set_layerwise_lr(cfg, model, epoch)
            experiment.n_query_init = full_n_query
            stem = str(_get(cfg, "file_name", "aline.pth")).split(".")[0]
            if rank == 0:
                say(f"burn-in finished; model saved at {save_state_dict(model, _get(cfg, 'output_dir', '.'), stem + '_burning.pth')}")
            barrier()
        optimizer_step(model, optimizer, burn_in=epoch < burn)
        scheduler.step()
        rec = dict(epoch=epoch, T=T, mask_type=mask_type, loss=float(terms["loss"]),
                   design_loss=float(terms["design_loss"]), predict_loss=float(terms["predict_loss"]),
                   lr=[g["lr"] for g in optimizer.param_groups], seconds=time.time() - tic)
        records.append(rec)
        if epoch % verbose == 0:
            say(f"Epoch: {epoch}, loss: {rec['loss']:.4f}, T: {T}, likelihood: {-rec['predict_loss']}, "
                f"design_loss: {rec['design_loss']}, predict_loss: {rec['predict_loss']}")
        if on_epoch is not None:
            on_epoch(rec)
        every = _get(cfg, "checkpoint", 0)
        if every and (epoch + 1) % every == 0:
            if rank == 0:
                save_checkpoint(cfg, model, optimizer, scheduler, epoch + 1, with_epoch=True)
            barrier()
    check_range_async(block=True)            # (rollouts posted by callers of train_step outside this loop)
    return records

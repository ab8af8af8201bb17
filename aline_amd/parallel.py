"""Episode-level data parallelism: episodes are independent through the whole
rollout, so every rank rolls out its own episodes and never exchanges data on the forward path.  One process per
GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on the GPU node, "gloo" in CPU tests).  bench.py times every leg
through `aggregate_throughput` (the whole-job rate of the bench contract).  The gradient all-reduce of the training step is
`aline_amd.train.all_reduce_grads`; the evaluation shards its outer batches in `aline_amd.utils.eval.eval_boed`."""
import os

import torch


def world_info():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def aggregate_throughput(local_units: float, local_seconds: float, dist=None, device="cpu", always=False):
    """Whole-job rate = units of all ranks / max-over-ranks time (the bench.py contract).  always: reduce also in a world of one
    (bench.py's single-GPU rehearsal of the RCCL path)."""
    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not always):
        return local_units / local_seconds, local_seconds, local_units
    t = torch.tensor([local_seconds], dtype=torch.float64, device=device)
    u = torch.tensor([local_units], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(u.item()) / float(t.item()), float(t.item()), float(u.item())

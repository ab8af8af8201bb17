"""Probe for DESIGN 4.3's note: does the streamed block-kernel path (ALINE_DBG_WIDE_BLOCKS) give bit-identical results run
to run with the library given in ALINE_HIP_LIB (built with / without -DALINE_RELU_INT)?  Prints the number of runs whose
log-probabilities / log-likelihoods differ from the first run, and which steps / episodes differ."""
import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, os.path.join(R, "oracle"))
from helpers import native_model
from aline_amd import _lib
from aline_amd.rollout import Rollout
_lib.lib.aline_debug_set_flags(_lib.DBG["WIDE_BLOCKS"])
from aline_amd.tasks import HiddenLocation
DIMS = {"dim_x": 2, "dim_y": 1, "d": 256, "F": 1024, "n_head": 8, "L": 2, "C": 10, "n_theta": 2, "embedding_type": "theta", "time_token": False}
model, _ = native_model(DIMS, 11, "bf16")
torch.manual_seed(5)
B, nq, T = 3, 200, 6
batch = HiddenLocation(device=torch.device("cuda"), n_query_init=nq).sample_batch(B)
g = torch.Generator().manual_seed(5)
forced = torch.stack([torch.stack([torch.randint(0, nq - t, (1,), generator=g)[0] for t in range(T)]) for _ in range(B)]).cuda()
import ctypes as C
from aline_amd import _lib
first, bad = None, []
logit0, rows_bad = None, {}
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    ro = Rollout(model, batch, T, select="forced", forced_idx=forced).run(); torch.cuda.synchronize()
    cur = (ro.log_prob.clone(), ro.target_ll.clone())
    off = _lib.lib.aline_debug_wlog_offset(C.byref(ro.m), C.byref(ro.r))
    lg = ro.ws[off:off + 4 * B * (1 + nq + 2)].view(torch.float32).clone()      # logits of the LAST step
    if logit0 is None:
        logit0 = lg
    else:
        for r in (lg != logit0).nonzero().flatten().tolist():
            rows_bad[r] = rows_bad.get(r, 0) + 1
    if first is None:
        first = cur
    elif not (torch.equal(cur[0], first[0]) and torch.equal(cur[1], first[1])):
        bad.append((i, (cur[0] != first[0]).nonzero().tolist()[:4], float((cur[0] - first[0]).abs().max())))
print(os.environ.get("ALINE_HIP_LIB", "default library"), ": runs differing from the first:", len(bad), bad[:3])
rows = sorted(rows_bad)
print("last-step logit rows (of", B * (1 + nq + 2), ") that ever differ:", len(rows), "tiles:", sorted({r // 16 for r in rows})[:40],
      "rows in tile:", sorted({r % 16 for r in rows}))

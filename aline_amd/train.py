"""Training objective of the reference (train_aline.py:80-152) on the native kernels: fused forward
rollout, REINFORCE reductions (tiny on-device torch ops), native backward of all T steps, and the
episode-data-parallel gradient all-reduce (one flat bucket per optimiser step)."""
import ctypes as C

import torch

from . import _lib
from .rollout import Rollout


def _grad_struct(model, into=None):
    """`aline_grads` pointing at param.grad of every weight (allocated + zeroed if missing), or at the
    tensors of `into` (dict id(param) -> tensor) when given."""
    if into is None:
        for p in model.parameters():
            if p.grad is None:
                p.grad = torch.zeros_like(p)

    def gp(p):
        return (p.grad if into is None else into[id(p)]).data_ptr()

    g = _lib.AlineGrads()
    emb, enc, head = model.embedder, model.encoder, model.head
    for nm in ("x", "y"):
        seq = getattr(emb, f"{nm}_embedder")
        setattr(g, f"{nm}_w1", gp(seq[0].weight))
        setattr(g, f"{nm}_b1", gp(seq[0].bias))
        setattr(g, f"{nm}_w2", gp(seq[2].weight))
        setattr(g, f"{nm}_b2", gp(seq[2].bias))
    if hasattr(emb, "theta_tokens"):
        g.theta_tokens = gp(emb.theta_tokens)
    for l, layer in enumerate(enc.encoder.layers):
        g.in_proj_w[l], g.in_proj_b[l] = gp(layer.self_attn.in_proj_weight), gp(layer.self_attn.in_proj_bias)
        g.out_proj_w[l], g.out_proj_b[l] = gp(layer.self_attn.out_proj.weight), gp(layer.self_attn.out_proj.bias)
        g.lin1_w[l], g.lin1_b[l] = gp(layer.linear1.weight), gp(layer.linear1.bias)
        g.lin2_w[l], g.lin2_b[l] = gp(layer.linear2.weight), gp(layer.linear2.bias)
        g.norm1_w[l], g.norm1_b[l] = gp(layer.norm1.weight), gp(layer.norm1.bias)
        g.norm2_w[l], g.norm2_b[l] = gp(layer.norm2.weight), gp(layer.norm2.bias)
    pr = head.acquisition_head.predictor
    g.acq_w1, g.acq_b1, g.acq_w2, g.acq_b2 = gp(pr[0].weight), gp(pr[0].bias), gp(pr[2].weight), gp(pr[2].bias)
    for c, h in enumerate(head.target_head.heads):
        g.gmm_w1[c], g.gmm_b1[c], g.gmm_w2[c], g.gmm_b2[c] = gp(h[0].weight), gp(h[0].bias), gp(h[2].weight), gp(h[2].bias)
    return g


_bwd_ws = _lib.Workspace()


def flat_grads(model):
    """One flat fp32 buffer that holds every gradient of `model`; `param.grad` are views into it.  The training step then
    zeroes, all-reduces and clips the gradients with one kernel each, and the `aline_grads` pointer table is built once
    (137 tensors at the default model: per-tensor zeroing, `clip_grad_norm_` and the ctypes table cost ~1 ms of host time per
    step, visible as GPU idle gaps).  Returns (flat, grads_struct); rebuilt when somebody replaced a `.grad`."""
    params = list(model.parameters())
    cache = getattr(model, "_aline_flat", None)
    if cache is not None:
        flat, ptrs, struct = cache
        try:
            if [p.grad.data_ptr() for p in params] == ptrs:
                return flat, struct
        except AttributeError:                      # a .grad is None
            pass
    total = sum(p.numel() for p in params)
    # (one extra element behind the gradients: the f16 range status of the step's rollout rides the gradient all-reduce in it,
    #  so every rank of a data-parallel job learns of an overflow on any rank from the collective it already makes)
    store = torch.zeros(total + 1, device=params[0].device, dtype=torch.float32)
    flat = store[:total]
    off = 0
    for p in params:
        n = p.numel()
        view = flat[off:off + n].view_as(p)
        if p.grad is not None:
            view.copy_(p.grad)
        p.grad = view
        off += n
    struct = _grad_struct(model)
    model._aline_flat = (flat, [p.grad.data_ptr() for p in params], struct)
    model._aline_flat_store = store
    return flat, struct


_DISC = {}


def _discounts(gamma, T, dev):
    """[gamma^1 .. gamma^(T-1)] on the device, built once per (gamma, T): a host list -> device tensor every step is a blocking
    copy in the middle of the step."""
    key = (float(gamma), int(T), str(dev))
    if key not in _DISC:
        _DISC[key] = torch.tensor([gamma ** t for t in range(1, T)], device=dev)
    return _DISC[key]


def reinforce_terms(ro, embedding_type, mask_type="all", gamma=1.0, alpha=1.0, burn_in=False, dist=None, world=1):
    """train_aline.py:97-132 on the rollout's per-step log-likelihoods.  Returns the losses and the two
    upstream gradients of `aline_rollout_backward` (dLoss/dlog_prob [B,T], dLoss/dtarget_ll [T,B,n_t]).

    dist / world (data parallel): the reward z-score of train_aline.py:122 uses the mean and the unbiased std over dim 0 of
    the GLOBAL batch (world * B episodes): one all-reduce of the per-rank sums [3, T - 1] (count, sum, sum of squares), so an
    N-rank step equals the single-process step on the concatenated batch; the design loss / its gradient are then scaled for
    the global batch (the gradient all-reduce averages over ranks).  dist=None: rank-local moments."""
    B, T, n_t = ro.B, ro.T, ro.n_t
    nll_q, nll = ro.nlls(embedding_type, mask_type)                      # [B, T]
    predict_loss = nll.mean()
    dev = nll.device
    g_logp = torch.zeros(B, T, device=dev)
    design_loss = torch.zeros((), device=dev)
    R = None
    if T > 1:
        disc = _discounts(gamma, T, dev)
        R = disc * torch.clamp(nll_q[:, :-1] - nll_q[:, 1:], min=0.0)    # [B, T-1], detached by construction
        if dist is not None and world > 1:
            mom = torch.stack([torch.full_like(R[0], float(B)), R.sum(0), (R * R).sum(0)])     # [3, T-1]
            dist.all_reduce(mom)
            n, mean = mom[0], mom[1] / mom[0]
            var = (mom[2] - n * mean * mean).clamp(min=0.0) / (n - 1.0)                         # unbiased, as Tensor.std
            R = (R - mean) / (var.sqrt() + 1e-9)
        else:
            R = (R - R.mean(dim=0, keepdim=True)) / (R.std(dim=0, keepdim=True) + 1e-9)
        design_loss = -(ro.log_prob[:, :-1] * R).mean()
        if not burn_in:
            g_logp[:, :-1] = -alpha * R / (B * (T - 1))
    n_th = ro.m.n_theta
    g_ll = torch.empty(T, B, n_t, device=dev)
    if embedding_type == "mix":                                           # two means (train_aline.py:106-107)
        n_td = n_t - n_th
        g_ll[..., :n_td] = -1.0 / (T * B * n_td)
        g_ll[..., n_td:] = -1.0 / (T * B * n_th)
    else:
        g_ll.fill_(-1.0 / (T * B * n_t))
    loss = predict_loss if burn_in else alpha * design_loss + predict_loss
    return dict(loss=loss, design_loss=design_loss, predict_loss=predict_loss, g_logp=g_logp.contiguous(),
                g_ll=g_ll.contiguous(), R=R)


def _default_bwd_ws_bytes(device):
    """Workspace budget of the per-op backward (which chunks the T steps to fit it): half of the card's memory -- 144 GB of the 288 GB
    of an MI355X (the d = 256 headline step then runs 10 steps per chunk instead of 2: fewer, longer launches) -- overridable by
    ALINE_BWD_WS_GB."""
    import os
    gb = os.environ.get("ALINE_BWD_WS_GB")
    if gb:
        return int(float(gb) * (1 << 30))
    try:
        free, total = torch.cuda.mem_get_info(device)
        held = torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)      # (cached by torch: reusable)
        return int(max(min(0.5 * total, 0.8 * (free + held)), 8 << 30))      # (ranks sharing a card in a rehearsal see less free memory)
    except Exception:
        return 24 << 30


def backward(model, ro, g_logp, g_ll, t_chunk=None, max_ws_bytes=None, grads=None):
    """Accumulates dLoss/dW into param.grad of every weight of `model` (C ABI aline_rollout_backward)."""
    if max_ws_bytes is None:
        max_ws_bytes = _default_bwd_ws_bytes(ro.device)
    m, r = ro.m, ro.r
    if grads is None:
        grads = _grad_struct(model)
    stale = ro.saved_acts is not None and not ro.saved_acts_valid()
    kept = r.saved_acts
    if stale:
        # the buffer was written by another rollout since this one's forward (a stale `ro`, or a shared buffer), or the library's
        # diagnostic word changed between forward and backward: the kept activations are not this rollout's -- recompute the layers
        r.saved_acts = None
    try:
        if t_chunk is None:
            t_chunk = ro.T
            while t_chunk > 1 and _lib.lib.aline_rollout_backward_workspace_bytes(C.byref(m), C.byref(r), t_chunk) > max_ws_bytes:
                t_chunk = (t_chunk + 1) // 2
        nbytes = _lib.lib.aline_rollout_backward_workspace_bytes(C.byref(m), C.byref(r), t_chunk)
        if nbytes == 0:
            raise RuntimeError("aline_amd: unsupported configuration for backward")
        ws = _bwd_ws.get(nbytes, ro.device)
        _lib.check(_lib.lib.aline_rollout_backward(C.byref(m), C.byref(r), g_logp.data_ptr(), g_ll.data_ptr(),
                                                   C.byref(grads), t_chunk, ws.data_ptr(), ws.numel(),
                                                   _lib.stream_ptr(ro.device)), "rollout_backward")
    finally:
        if stale:
            r.saved_acts = kept
    return t_chunk


ALLREDUCE_CALLS = 0      # collectives issued by this process (bench.py reports the count per optimiser step)


def all_reduce_grads(model, dist, world, flat=None):
    """One flat-bucket all-reduce (sum, then / world) of every gradient (RCCL over xGMI on the node).  `flat`: the
    buffer of `flat_grads` (the gradients are views of it: no gather / scatter copies)."""
    global ALLREDUCE_CALLS
    ALLREDUCE_CALLS += 1
    if flat is not None:
        store = getattr(model, "_aline_flat_store", None)
        buf = store if (store is not None and store.data_ptr() == flat.data_ptr()) else flat
        dist.all_reduce(buf)           # gradients + the range-status slot behind them: ONE collective
        flat /= world
        return
    params = [p for p in model.parameters() if p.grad is not None]
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    dist.all_reduce(flat)
    flat /= world
    off = 0
    for p in params:
        n = p.numel()
        p.grad.copy_(flat[off:off + n].view_as(p))
        off += n


_SAVED_ACTS = {}              # device -> one grow-only buffer for the activations the training rollouts keep for their backward


def _shared_acts(nbytes, device=None):
    """The shared `saved_acts` buffer of `device` (Rollout(keep_acts=...) passes the rollout's device).  Growing it invalidates the
    captured rollouts of that device, whose graphs hold the old address: they are dropped from the cache."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    buf = _SAVED_ACTS.get(dev)
    if buf is None or buf.numel() * 4 < nbytes:
        for k in [k for k, ro in _ROLLOUT_GRAPHS.items() if ro.device == dev]:
            del _ROLLOUT_GRAPHS[k]
        _SAVED_ACTS.pop(dev, None)
        buf = None                                   # (free the old one before allocating the new)
        _SAVED_ACTS[dev] = buf = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=dev)
    return buf


_ROLLOUT_GRAPHS = {}          # (model, shapes, T) -> captured Rollout, most recently used last; at most _MAX_GRAPHS of them
_MAX_GRAPHS = 8
_GRAPH_OK = True
_GRAPH_MISSES = 0             # consecutive cache misses: a driver that draws T from a wide range would capture every step
_RANGE_PENDING = []           # (pinned status word, event) of training rollouts whose f16 range status has not been looked at


def _own(t):
    """A private fp32 copy: the graph's input buffers must not alias the caller's first batch (`_lib.f32` returns the same
    storage for an fp32-contiguous tensor, and the buffers are overwritten at every later step)."""
    return None if t is None else t.detach().to(torch.float32).clone()


def check_range_async(block=False):
    """Training rollouts in f16x3 post their range status (include/aline_hip.h) to pinned memory without stalling the step;
    this looks at the ones that have arrived (all of them with block=True) and raises if an operand left f16's range -- the
    update of that step was computed from inf / NaN, so the run must stop (use precision 'f32')."""
    keep = []
    for word, ev in _RANGE_PENDING:
        if block:
            ev.synchronize()
        if ev.query():
            if int(word.item()) & 0xFF:
                _RANGE_PENDING.clear()
                raise RuntimeError("aline_amd: an F16X3 operand left f16's range during a training rollout (status "
                                   f"{int(word.item()) & 0xFF}); train with precision 'f32'")
        else:
            keep.append((word, ev))
    _RANGE_PENDING[:] = keep


_RANGE_WORDS = None           # ring of pinned status words (pinning per step would cost a host allocation each time)
_RANGE_NEXT = 0


def _post_range_status(ro):
    global _RANGE_WORDS, _RANGE_NEXT
    if ro.m.precision != _lib.PREC["f16x3"]:
        return
    if _RANGE_WORDS is None:
        _RANGE_WORDS = torch.zeros(64, dtype=torch.int32).pin_memory()
    if len(_RANGE_PENDING) >= 48:          # (nobody looked for 48 steps: look now rather than wrap the ring)
        check_range_async(block=True)
    word = _RANGE_WORDS[_RANGE_NEXT:_RANGE_NEXT + 1]
    _RANGE_NEXT = (_RANGE_NEXT + 1) % 64
    off = _lib.lib.aline_f16_range_offset()
    word.copy_(ro.ws[off:off + 4].view(torch.int32), non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    _RANGE_PENDING.append((word, ev))


def _sampled_rollout(model, batch, T):
    """The forward rollout of a training step (designs sampled, train_aline.py:80-110) replayed from a HIP graph that is kept
    per (model, batch shapes, T): the step's data is copied into the graph's own input buffers, the uniform numbers are redrawn
    and the graph is replayed (3.2 ms of eager launches -> 2.9 ms at the headline shape).  ALINE_TRAIN_GRAPH=0: eager.  The
    cache is LRU; after _MAX_GRAPHS consecutive misses (T drawn from a range wider than the cache, train_aline.py:59-62) the
    step runs eagerly instead of capturing a graph it will not see again."""
    import os
    global _GRAPH_OK, _GRAPH_MISSES
    if os.environ.get("ALINE_TRAIN_GRAPH", "1") == "0" or not _GRAPH_OK:
        return Rollout(model, batch, T, select="sample", keep_acts=_shared_acts).run()
    g = lambda k: batch.get(k) if isinstance(batch, dict) else getattr(batch, k, None)    # noqa: E731
    tens = {k: g(k) for k in ("context_x", "context_y", "query_x", "query_y", "target_all", "target_x", "target_mask")}
    prm = list(model.parameters())
    key = (id(model), prm[0].data_ptr(), prm[-1].data_ptr(), T, model.precision if hasattr(model, "precision") else None,
           tuple((k, tuple(v.shape), v.dtype) for k, v in tens.items() if torch.is_tensor(v)))
    ro = _ROLLOUT_GRAPHS.pop(key, None)
    if ro is None:
        _GRAPH_MISSES += 1
        if _GRAPH_MISSES > _MAX_GRAPHS:          # thrashing: eager launches (+0.3 ms) beat a warm-up run + a capture per step
            return Rollout(model, batch, T, select="sample", keep_acts=_shared_acts).run()
        if len(_ROLLOUT_GRAPHS) >= _MAX_GRAPHS:
            _ROLLOUT_GRAPHS.pop(next(iter(_ROLLOUT_GRAPHS)))         # least recently used
        try:
            own = dict(batch) if isinstance(batch, dict) else {k: v for k, v in vars(batch).items()}
            for k in ("target_all", "target_x"):
                if torch.is_tensor(own.get(k)):
                    own[k] = _own(own[k])
            if torch.is_tensor(own.get("target_mask")):
                own["target_mask"] = own["target_mask"].clone()
            ro = Rollout(model, own, T, select="sample", keep_acts=_shared_acts).capture()
        except RuntimeError as e:                   # a runtime that refuses the capture: eager launches from now on
            import warnings
            warnings.warn(f"aline_amd: HIP-graph capture of the training rollout failed ({e}); using eager launches")
            _GRAPH_OK = False
            torch.cuda.synchronize()
            return Rollout(model, batch, T, select="sample", keep_acts=_shared_acts).run()
    else:
        _GRAPH_MISSES = 0
        torch.cat([_lib.f32(tens["context_x"]), _lib.f32(tens["query_x"])], dim=1, out=ro.px)
        torch.cat([_lib.f32(tens["context_y"]), _lib.f32(tens["query_y"])], dim=1, out=ro.py)
        ro.target_all.copy_(tens["target_all"].reshape(ro.target_all.shape))
        if ro.tx is not None:
            ro.tx.copy_(tens["target_x"])
        if ro.tmask is not None:
            ro.tmask.copy_(tens["target_mask"])
    _ROLLOUT_GRAPHS[key] = ro                        # (re-)inserted last = most recently used
    ro.refresh_uniform()
    return ro.replay()


def optimizer_step(model, optimizer, burn_in=False):
    """optimizer.step() as the reference sees it: during burn-in (train_aline.py:126-128: loss = predict_loss) no loss term
    reaches the acquisition head, its .grad stays None there and AdamW skips those parameters -- no weight decay, no moment
    updates.  The flat gradient buffer gives every parameter a (zero) .grad, so they are hidden for the step."""
    if not burn_in:
        optimizer.step()
        return
    frozen = list(model.head.acquisition_head.parameters())
    views = [p.grad for p in frozen]
    for p in frozen:
        p.grad = None
    try:
        optimizer.step()
    finally:
        for p, v in zip(frozen, views):
            p.grad = v


def train_step(model, batch, T, optimizer=None, embedding_type="theta", mask_type="all", gamma=1.0, alpha=1.0,
               burn_in=False, forced_idx=None, clip_grads=True, dist=None, world=1, t_chunk=None, global_reward_moments=False,
               force_collective=False):
    """One epoch body of train_aline.py:55-152 (without the hydra / logging shell).

    The returned `ro` is the step's rollout; when it came from the graph cache its outputs (log_prob, target_ll, idx, ...) are
    the cache entry's buffers and are overwritten by the next step with the same shapes -- clone what must outlive the step.
    The activations it kept for the backward (`ro.saved_acts`) live in ONE buffer shared by all training rollouts of the process and
    are overwritten by the next step's rollout whatever its shape.
    global_reward_moments (N > 1): z-score the rewards with the moments of the WHOLE data-parallel batch (one extra all-reduce
    of [3, T - 1] floats before the backward; SURVEY 8-e option ii) instead of the rank-local moments (option i).
    force_collective: issue the gradient all-reduce also at world == 1 (bench.py's single-GPU rehearsal of the RCCL path)."""
    model.train()
    check_range_async()                   # (f16x3 rollouts of earlier steps: raises if an operand left f16's range)
    with torch.no_grad():
        flat, gstruct = flat_grads(model)
        flat.zero_()
        if forced_idx is not None:
            ro = Rollout(model, batch, T, select="forced", forced_idx=forced_idx, keep_acts=_shared_acts).run()
        else:
            ro = _sampled_rollout(model, batch, T)
        _post_range_status(ro)
        store = getattr(model, "_aline_flat_store", None)
        status = None
        if store is not None and store.data_ptr() == flat.data_ptr():
            status = store[-1:]                      # rides the gradient all-reduce (all_reduce_grads): > 0 iff ANY rank overflowed
            if ro.m.precision == _lib.PREC["f16x3"]:
                off = _lib.lib.aline_f16_range_offset()
                status.copy_((ro.ws[off:off + 4].view(torch.int32) & 0xFF).ne(0).to(torch.float32))
            else:
                status.zero_()
        terms = reinforce_terms(ro, embedding_type, mask_type, gamma, alpha, burn_in,
                                dist=dist if (global_reward_moments and world > 1) else None, world=world)
        backward(model, ro, terms["g_logp"], terms["g_ll"], t_chunk=t_chunk, grads=gstruct)
        if status is not None and ro.m.precision == _lib.PREC["f16x3"]:
            # the f16 gradient products of the backward scale their operands by the power of two of the upstream gradient's maximum; if a
            # tile program's own growth still left f16's range the gradients are inf / NaN, never silently wrong: same slot, same refusal
            status.add_((~torch.isfinite(flat.abs().max())).to(torch.float32))
        if dist is not None and (world > 1 or force_collective):
            all_reduce_grads(model, dist, world, flat=flat)
        if clip_grads:
            # torch.nn.utils.clip_grad_norm_(parameters, max_norm=1.0, norm_type="inf") of train_aline.py:138 on the flat buffer:
            # total norm = max |g| over all tensors, coefficient = min(1, max_norm / (total + 1e-6))
            flat.mul_((1.0 / (flat.abs().max() + 1e-6)).clamp(max=1.0))
        terms["range_status"] = status           # device tensor [1] (or None): see check_training_range
        if optimizer is not None:
            optimizer_step(model, optimizer, burn_in)
    return terms, ro


def check_training_range(terms):
    """Raises -- on EVERY rank of a data-parallel job together -- if the rollout of the training step that produced `terms` left
    f16's range on any rank: the status slot behind the flat gradient buffer went through the step's gradient all-reduce, so all
    ranks read the same value.  Host-synchronising (4 bytes): call it where the loop synchronises anyway, BEFORE the optimiser step
    and before a checkpoint is written (the gradients of such a step are inf / NaN)."""
    st = terms.get("range_status")
    if st is not None and float(st) > 0:
        _RANGE_PENDING.clear()
        raise RuntimeError("aline_amd: an F16X3 operand left f16's range during a training rollout or its backward (on this or another rank); "
                           "the step's gradients are not applied -- train with precision 'f32'")


def step_backward(model, batch, idx, g_logp, g_mean=None, g_std=None, g_weight=None):
    """Gradients of one `Aline.forward(batch)` call (the reference's per-step API) wrt every parameter,
    given the upstream gradients of design_out.log_prob [B] and posterior_out.mixture_* [B,n_t,C].
    The step is expressed in the static-slot layout (T = 1) and handed to aline_rollout_backward_ex."""
    from .model import _native
    g = _native._get
    m = model.model_struct()
    cx, cy, qx = _lib.f32(g(batch, "context_x")), _lib.f32(g(batch, "context_y")), _lib.f32(g(batch, "query_x"))
    dev = cx.device
    B, n_c, n_q = cx.shape[0], cx.shape[1], qx.shape[1]
    P = n_c + n_q
    px = torch.cat([cx, qx], dim=1).contiguous()
    py = torch.cat([cy, torch.zeros(B, n_q, cy.shape[-1], device=dev)], dim=1).contiguous()
    role = torch.zeros(B, P, dtype=torch.int32, device=dev)
    role[:, :n_c] = torch.arange(1, n_c + 1, dtype=torch.int32, device=dev)
    ta = g(batch, "target_all")
    n_t = ta.shape[1]
    target_all = _lib.f32(ta.reshape(B, n_t))
    n_td = n_t - m.n_theta
    tx = g(batch, "target_x")
    tx = _lib.f32(tx) if (tx is not None and n_td > 0) else None
    tm = g(batch, "target_mask")
    tmask = None if tm is None else tm.to(dev, torch.uint8).contiguous()
    slot = (n_c + idx.reshape(B).to(torch.int32)).contiguous()
    r = _lib.AlineRollout()
    r.B, r.P, r.n_ctx0, r.n_target_data, r.T = B, P, n_c, n_td, 1
    r.point_x, r.point_y, r.role = px.data_ptr(), py.data_ptr(), role.data_ptr()
    r.target_x, r.target_all, r.target_mask = _lib.ptr(tx), target_all.data_ptr(), _lib.ptr(tmask)
    r.slot = slot.data_ptr()
    params = list(model.parameters())
    into = {id(p): torch.zeros_like(p) for p in params}
    grads = _grad_struct(model, into)
    glp = _lib.f32(g_logp.reshape(B, 1)) if g_logp is not None else torch.zeros(B, 1, device=dev)
    gm, gs, gw = (None if t is None else _lib.f32(t.reshape(1, B, n_t, -1)) for t in (g_mean, g_std, g_weight))
    if gm is None and gs is None and gw is None:
        gm = torch.zeros(1, B, n_t, m.C, device=dev)
    nbytes = _lib.lib.aline_rollout_backward_workspace_bytes(C.byref(m), C.byref(r), 1)
    if nbytes == 0:
        raise RuntimeError("aline_amd: unsupported configuration for backward")
    ws = _bwd_ws.get(nbytes, dev)
    _lib.check(_lib.lib.aline_rollout_backward_ex(C.byref(m), C.byref(r), glp.data_ptr(), None, _lib.ptr(gm),
                                                  _lib.ptr(gs), _lib.ptr(gw), C.byref(grads), 1, ws.data_ptr(),
                                                  ws.numel(), _lib.stream_ptr(dev)), "rollout_backward_ex")
    return [into[id(p)] for p in params]

"""T-step acquisition loop (reference train_aline.py:80-110 / utils/eval.py:24-30) on the
shape-static slot layout of the C ABI (`aline_rollout_*`): the whole loop -- forward, design
selection, context/query update, GMM log-likelihood -- stays on the device with no host sync, and
can be captured in one HIP graph."""
import ctypes as C

import torch

from . import _lib
from .model import _native


_ACTS_WRITES = {}        # saved_acts buffer address -> number of forwards that wrote it (Rollout._stamp_acts)


class Rollout:
    """One batch of episodes rolled out for T design steps.

    select: 'argmax' (eval, head.py:355-358), 'sample' (train, head.py:350-354; randoms from
    `uniform` [T, B] or torch.rand), or 'forced' with `forced_idx` [B, T] (teacher forcing).
    """

    def __init__(self, model, batch, T, select="argmax", forced_idx=None, uniform=None,
                 time_token_T=0, keep_zt=False, keep_posterior=True, time_token_reverse=False,
                 keep_query_posterior=False, keep_acts=False):
        self.model = model
        self.m = model.model_struct()
        g = _native._get
        cx, cy = _lib.f32(g(batch, "context_x")), _lib.f32(g(batch, "context_y"))
        qx, qy = _lib.f32(g(batch, "query_x")), _lib.f32(g(batch, "query_y"))
        if not cx.is_cuda:
            raise RuntimeError("aline_amd: batch tensors must live on the GPU (no CPU fallback)")
        dev = self.device = cx.device
        B, n_c0, n_q0 = cx.shape[0], cx.shape[1], qx.shape[1]
        self.B, self.n_c0, self.n_q0, self.T = B, n_c0, n_q0, T
        self.P = P = n_c0 + n_q0
        self.px = torch.cat([cx, qx], dim=1).contiguous()
        self.py = torch.cat([cy, qy], dim=1).contiguous()
        ta = g(batch, "target_all")
        self.n_t = n_t = ta.shape[1]
        self.target_all = _lib.f32(ta.reshape(B, n_t))
        n_td = n_t - self.m.n_theta
        tx = g(batch, "target_x")
        self.tx = _lib.f32(tx) if (tx is not None and n_td > 0) else None
        tm = g(batch, "target_mask")
        self.tmask = None if tm is None else tm.to(dev, torch.uint8).contiguous()
        C_ = self.m.C
        self.role = torch.empty(B, P, dtype=torch.int32, device=dev)
        self.idx = torch.empty(B, T, dtype=torch.int64, device=dev)
        self.slot = torch.empty(B, T, dtype=torch.int32, device=dev)
        self.log_prob = torch.empty(B, T, device=dev)
        self.target_ll = torch.empty(T, B, n_t, device=dev)
        self.zt = torch.empty(T, B, n_q0, device=dev) if keep_zt else None
        if keep_posterior:
            self.post_mean = torch.empty(T, B, n_t, C_, device=dev)
            self.post_std = torch.empty_like(self.post_mean)
            self.post_weight = torch.empty_like(self.post_mean)
        else:
            self.post_mean = self.post_std = self.post_weight = None
        # posterior_out_query of every step, by slot (model/head.py:366; lazy in the step API, optional here)
        self.postq_mean = self.postq_std = self.postq_weight = None
        if keep_query_posterior:
            self.postq_mean = torch.empty(T, B, P, C_, device=dev)
            self.postq_std = torch.empty_like(self.postq_mean)
            self.postq_weight = torch.empty_like(self.postq_mean)
        r = self.r = _lib.AlineRollout()
        r.B, r.P, r.n_ctx0, r.n_target_data, r.T = B, P, n_c0, n_td, T
        r.point_x, r.point_y, r.role = self.px.data_ptr(), self.py.data_ptr(), self.role.data_ptr()
        r.target_x = _lib.ptr(self.tx)
        r.target_all = self.target_all.data_ptr()
        r.target_mask = _lib.ptr(self.tmask)
        r.select_mode = {"argmax": _lib.SELECT_ARGMAX, "sample": _lib.SELECT_SAMPLE,
                         "forced": _lib.SELECT_FORCED}[select]
        self.uniform = self.forced = None
        if select == "sample":
            self.uniform = (uniform if uniform is not None else torch.rand(T, B, device=dev))
            self.uniform = self.uniform.to(dev, torch.float32).contiguous()
            r.uniform = self.uniform.data_ptr()
        if select == "forced":
            self.forced = forced_idx.to(dev, torch.int64).contiguous()
            assert self.forced.shape == (B, T)
            r.forced_idx = self.forced.data_ptr()
        # step t feeds t / T (training loop) or, reversed, (T - t) / T (the reference's eval loop, utils/eval.py:24)
        r.time_token_T = -(time_token_T or T) if time_token_reverse else time_token_T
        r.idx, r.slot, r.log_prob = self.idx.data_ptr(), self.slot.data_ptr(), self.log_prob.data_ptr()
        r.target_ll = self.target_ll.data_ptr()
        r.zt = _lib.ptr(self.zt)
        r.post_mean, r.post_std, r.post_weight = (_lib.ptr(self.post_mean), _lib.ptr(self.post_std),
                                                  _lib.ptr(self.post_weight))
        r.postq_mean, r.postq_std, r.postq_weight = (_lib.ptr(self.postq_mean), _lib.ptr(self.postq_std),
                                                     _lib.ptr(self.postq_weight))
        # training rollouts: the encoder layers' inputs and attention outputs of every (step, episode, row), kept for the backward
        # (aline_rollout.saved_acts: the s3 path writes them, aline_rollout_backward reads them instead of recomputing the layers)
        # keep_acts: True (own buffer) or a callable nbytes -> float32 tensor of at least that size (a buffer shared by the cached
        # rollouts of a training loop: only one rollout is between its forward and its backward at a time)
        self.saved_acts = None
        if keep_acts:
            sb = _lib.lib.aline_rollout_saved_acts_bytes(C.byref(self.m), C.byref(r))
            if sb:
                self.saved_acts = keep_acts(sb, dev) if callable(keep_acts) else torch.empty(sb // 4, dtype=torch.float32, device=dev)
                assert self.saved_acts.numel() * 4 >= sb and self.saved_acts.device == dev
                r.saved_acts = self.saved_acts.data_ptr()
        nbytes = _lib.lib.aline_rollout_workspace_bytes(C.byref(self.m), C.byref(r))
        if nbytes == 0:
            raise RuntimeError("aline_amd: unsupported model/batch configuration")
        self.ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        self._graph = None
        self.fell_back = False          # run_checked() re-ran the rollout in f32 after an f16 range overflow
        self._acts_stamp = None         # (write generation of saved_acts, library diagnostic state) of this rollout's last forward

    PATHS = {0: "generic pipeline", 1: "fused::rollout_f32_kernel",
             3: "x3::layer_kernel", 4: "s3::step_kernel", 5: "x5::layer_kernel"}

    @property
    def path(self):
        """The implementation aline_rollout_forward picks for this model / batch (named by its dominant kernel)."""
        rc = _lib.lib.aline_rollout_path(C.byref(self.m), C.byref(self.r))
        if rc < 0:
            _lib.check(rc, "rollout_path")
        return self.PATHS[rc]

    @property
    def kernel_name(self):
        """The dominant kernel of that path as rocprofv3 names it (template arguments of the launch shape included)."""
        buf = C.create_string_buffer(128)
        rc = _lib.lib.aline_rollout_kernel_name(C.byref(self.m), C.byref(self.r), buf, 128)
        if rc < 0:
            _lib.check(rc, "rollout_kernel_name")
        return buf.value.decode()

    # ------------------------------------------------------------------------------------------
    def refresh_uniform(self):
        if self.uniform is not None:
            self.uniform.uniform_()

    def _stamp_acts(self):
        """Records that THIS rollout's forward is the last writer of its `saved_acts` buffer (which may be shared with other
        rollouts) and under which diagnostic state of the library it ran: `saved_acts_valid()` is what the backward asks before
        it reads the buffer instead of recomputing the layers."""
        if self.saved_acts is not None:
            key = self.saved_acts.data_ptr()
            gen = _ACTS_WRITES.get(key, 0) + 1
            _ACTS_WRITES[key] = gen
            self._acts_stamp = (gen, _lib.debug_state())

    def saved_acts_valid(self):
        """True iff `saved_acts` holds the activations of this rollout's last forward: nobody wrote the (shared) buffer since, and the
        library's diagnostic word (which decides whether the forward path writes them and the backward reads them) is unchanged."""
        if self.saved_acts is None or self._acts_stamp is None:
            return False
        gen, state = self._acts_stamp
        return _ACTS_WRITES.get(self.saved_acts.data_ptr()) == gen and state == _lib.debug_state()

    def _enqueue(self):
        st = _lib.stream_ptr(self.device)
        _lib.check(_lib.lib.aline_rollout_forward(C.byref(self.m), C.byref(self.r), self.ws.data_ptr(),
                                                  self.ws.numel(), st), "rollout_forward")
        self._stamp_acts()

    def run(self):
        """init + T steps enqueued on the current stream (no host synchronisation)."""
        self._enqueue()
        return self

    # ---- f16 range guard (include/aline_hip.h: aline_f16_range_status) -----------------------------------------------
    def range_status(self):
        """0 = clean; bit 0 / bit 1: an activation / a weight left f16's range in an F16X3 kernel of the last run (the
        results then hold inf / NaN).  Host-synchronising: call it where the caller synchronises anyway."""
        return _lib.f16_range_status(self.ws, self.device)

    def check_range(self):
        st = self.range_status()
        if st:
            raise RuntimeError(f"aline_amd: F16X3 operand out of f16 range (status {st}: "
                               f"{'activation ' if st & 1 else ''}{'weight' if st & 2 else ''}); use precision 'f32'")
        return self

    def run_checked(self):
        """run(), then the range status (one host synchronisation); an F16X3 rollout whose operands left f16's range is
        re-run in exact fp32 (same designs when they were forced; a warning is issued)."""
        self.run()
        if self.m.precision != _lib.PREC["f16x3"] or not self.range_status():
            return self
        import warnings
        warnings.warn("aline_amd: an F16X3 operand left f16's range (|x| >= 65504 or non-finite); re-running this rollout in f32")
        self.m.precision = _lib.PREC["f32"]
        nbytes = _lib.lib.aline_rollout_workspace_bytes(C.byref(self.m), C.byref(self.r))
        if nbytes == 0:
            raise RuntimeError("aline_amd: unsupported model/batch configuration")
        if nbytes > self.ws.numel():
            self.ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        self._graph = None
        self.fell_back = True
        return self.run()

    def capture(self):
        """Capture init + T steps into one HIP graph (hipGraph through torch's stream capture:
        the C ABI allocates nothing and only enqueues on the capturing stream)."""
        self._enqueue()                       # warm-up outside capture (function attributes etc.)
        torch.cuda.synchronize(self.device)
        g = torch.cuda.CUDAGraph()
        # thread_local: other threads of the process (the RCCL watchdog under torch.distributed polls events) may keep
        # calling the HIP runtime while this thread captures
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            self._enqueue()
        self._graph = g
        return self

    def replay(self):
        self._graph.replay()
        self._stamp_acts()
        return self

    def step(self, t):
        st = _lib.stream_ptr(self.device)
        if t == 0:
            _lib.check(_lib.lib.aline_rollout_init(C.byref(self.m), C.byref(self.r), self.ws.data_ptr(),
                                                   self.ws.numel(), st), "rollout_init")
        _lib.check(_lib.lib.aline_rollout_step(C.byref(self.m), C.byref(self.r), t, self.ws.data_ptr(),
                                               self.ws.numel(), st), "rollout_step")

    def export_context(self, n_ctx=None):
        """context_x / context_y in order of acquisition, as Task.update_batch would have built them
        (tasks/base_task.py:133-154)."""
        n_ctx = self.n_c0 + self.T if n_ctx is None else n_ctx
        dx, dy = self.px.shape[-1], self.py.shape[-1]
        cx = torch.empty(self.B, n_ctx, dx, device=self.device)
        cy = torch.empty(self.B, n_ctx, dy, device=self.device)
        _lib.check(_lib.lib.aline_rollout_export(C.byref(self.r), n_ctx, cx.data_ptr(), cy.data_ptr(),
                                                 None, None, dx, dy, _lib.stream_ptr(self.device)),
                   "rollout_export")
        return cx, cy

    # train_aline.py:97-110 reductions on the per-step log-likelihoods (tiny, torch on device)
    def nlls(self, embedding_type, mask_type="all"):
        ll = self.target_ll                                      # [T, B, n_t]
        n_th = self.m.n_theta
        if self.tmask is not None:
            sel = torch.where(self.tmask.bool())[0]
            masked = ll[:, :, sel]
        else:
            masked = ll
        if embedding_type == "mix" and mask_type == "all":
            nll_q = -(masked[..., :-n_th].mean(-1) + masked[..., -n_th:].mean(-1))
        else:
            nll_q = -masked.mean(-1)
        if embedding_type == "mix":
            nll = -(ll[..., :-n_th].mean(-1) + ll[..., -n_th:].mean(-1))
        else:
            nll = -ll.mean(-1)
        return nll_q.t(), nll.t()                                # [B, T]

"""ctypes binding of libaline_hip.so (the C ABI declared in include/aline_hip.h).

The product path has no CPU fallback: if the HIP library is missing, importing this module
raises, and every op that needs it fails loudly.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ALINE_HIP_LIB") or os.path.join(_HERE, "csrc", "libaline_hip.so")   # (override: kernel timing experiments)

MAX_LAYERS = 8
MAX_COMPONENTS = 16
EMB = {"data": 0, "theta": 1, "mix": 2}
PREC = {"f32": 0, "fp32": 0, "bf16": 1, "bf16x3": 2, "f16x3": 3}
SELECT_ARGMAX, SELECT_SAMPLE, SELECT_FORCED = 0, 1, 2

_fp = C.c_void_p  # device pointers travel as plain addresses


class AlineModel(C.Structure):
    _fields_ = (
        [(n, C.c_int32) for n in ("dim_x", "dim_y", "d", "F", "H", "L", "C", "n_theta",
                                  "embedding_type", "time_token")]
        + [("std_min", C.c_float), ("precision", C.c_int32)]
        + [(n, _fp) for n in ("x_w1", "x_b1", "x_w2", "x_b2", "y_w1", "y_b1", "y_w2", "y_b2",
                              "theta_tokens")]
        + [(n, _fp * MAX_LAYERS) for n in ("in_proj_w", "in_proj_b", "out_proj_w", "out_proj_b",
                                           "lin1_w", "lin1_b", "lin2_w", "lin2_b", "norm1_w",
                                           "norm1_b", "norm2_w", "norm2_b")]
        + [(n, _fp) for n in ("acq_w1", "acq_b1", "acq_w2", "acq_b2")]
        + [(n, _fp * MAX_COMPONENTS) for n in ("gmm_w1", "gmm_b1", "gmm_w2", "gmm_b2")]
    )


class AlineGrads(C.Structure):
    _fields_ = (
        [(n, _fp) for n in ("x_w1", "x_b1", "x_w2", "x_b2", "y_w1", "y_b1", "y_w2", "y_b2",
                            "theta_tokens")]
        + [(n, _fp * MAX_LAYERS) for n in ("in_proj_w", "in_proj_b", "out_proj_w", "out_proj_b",
                                           "lin1_w", "lin1_b", "lin2_w", "lin2_b", "norm1_w",
                                           "norm1_b", "norm2_w", "norm2_b")]
        + [(n, _fp) for n in ("acq_w1", "acq_b1", "acq_w2", "acq_b2")]
        + [(n, _fp * MAX_COMPONENTS) for n in ("gmm_w1", "gmm_b1", "gmm_w2", "gmm_b2")]
    )


class AlineStep(C.Structure):
    _fields_ = (
        [(n, C.c_int32) for n in ("B", "n_ctx", "n_query", "n_target_data")]
        + [(n, _fp) for n in ("context_x", "context_y", "query_x", "target_x", "target_all",
                              "target_mask", "time_t")]
        + [("select_mode", C.c_int32)]
        + [(n, _fp) for n in ("uniform", "forced_idx", "idx", "log_prob", "zt", "post_mean",
                              "post_std", "post_weight", "postq_mean", "postq_std", "postq_weight",
                              "target_ll", "embedding", "encoding")]
    )


class AlineRollout(C.Structure):
    _fields_ = (
        [(n, C.c_int32) for n in ("B", "P", "n_ctx0", "n_target_data", "T")]
        + [(n, _fp) for n in ("point_x", "point_y", "role", "target_x", "target_all", "target_mask")]
        + [("select_mode", C.c_int32)]
        + [(n, _fp) for n in ("uniform", "forced_idx")]
        + [("time_token_T", C.c_int32)]
        + [(n, _fp) for n in ("idx", "slot", "log_prob", "target_ll", "zt", "post_mean", "post_std",
                              "post_weight", "ev_kernel_start", "ev_kernel_stop", "postq_mean", "postq_std", "postq_weight", "saved_acts")]
        + [("ev_kernel_step", C.c_int32)]
    )


def _load():
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"aline_amd: HIP library not built ({LIB_PATH} missing). Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C aline_amd/csrc`. "
            "There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    MP, SP, RP = C.POINTER(AlineModel), C.POINTER(AlineStep), C.POINTER(AlineRollout)
    GP = C.POINTER(AlineGrads)
    sig = {
        "aline_abi_version": (C.c_int, []),
        "aline_error_string": (C.c_char_p, [C.c_int]),
        "aline_step_workspace_bytes": (C.c_size_t, [MP, SP]),
        "aline_embed_forward": (C.c_int, [MP, SP, _fp, C.c_size_t, _fp]),
        "aline_encoder_forward": (C.c_int, [MP, SP, _fp, _fp, C.c_size_t, _fp]),
        "aline_head_forward": (C.c_int, [MP, SP, _fp, _fp, C.c_size_t, _fp]),
        "aline_step_forward": (C.c_int, [MP, SP, _fp, C.c_size_t, _fp]),
        "aline_rollout_workspace_bytes": (C.c_size_t, [MP, RP]),
        "aline_rollout_init": (C.c_int, [MP, RP, _fp, C.c_size_t, _fp]),
        "aline_rollout_step": (C.c_int, [MP, RP, C.c_int, _fp, C.c_size_t, _fp]),
        "aline_rollout_forward": (C.c_int, [MP, RP, _fp, C.c_size_t, _fp]),
        "aline_rollout_path": (C.c_int, [MP, RP]),
        "aline_rollout_saved_acts_bytes": (C.c_size_t, [MP, RP]),
        "aline_rollout_kernel_name": (C.c_int, [MP, RP, C.c_char_p, C.c_size_t]),
        "aline_rollout_export": (C.c_int, [RP, C.c_int, _fp, _fp, _fp, _fp, C.c_int, C.c_int, _fp]),
        "aline_compute_ll": (C.c_int, [_fp, _fp, _fp, _fp, C.c_int64, C.c_int, _fp, _fp]),
        "aline_eig_location_step": (C.c_int, [_fp, _fp, _fp, _fp, C.c_int64, C.c_int, C.c_int,
                                              C.c_int, C.c_float, C.c_float, C.c_float, _fp]),
        "aline_eig_ces_step": (C.c_int, [_fp, _fp, _fp, _fp, C.c_int64, C.c_int, C.c_float,
                                         C.c_float, _fp, _fp]),
        "aline_eig_finalize_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int]),
        "aline_f16_range_offset": (C.c_size_t, []),
        "aline_f16_range_status": (C.c_int, [_fp, _fp]),
        "aline_debug_set_flags": (C.c_uint32, [C.c_uint32]),
        "aline_debug_get_flags": (C.c_uint32, []),
        "aline_debug_set_param": (C.c_int, [C.c_int, C.c_int]),
        "aline_debug_get_param": (C.c_int, [C.c_int]),
        "aline_debug_stamps_offset": (C.c_size_t, [MP, RP]),
        "aline_debug_xraw_offset": (C.c_size_t, [MP, RP]),
        "aline_cholesky_upper": (C.c_int, [_fp, C.c_int, C.c_int, _fp, _fp]),
        "aline_rollout_backward_workspace_bytes": (C.c_size_t, [MP, RP, C.c_int]),
        "aline_rollout_backward": (C.c_int, [MP, RP, _fp, _fp, GP, C.c_int, _fp, C.c_size_t, _fp]),
        "aline_rollout_backward_ex": (C.c_int, [MP, RP, _fp, _fp, _fp, _fp, _fp, GP, C.c_int, _fp, C.c_size_t, _fp]),
        "aline_eig_finalize": (C.c_int, [_fp, C.c_int64, C.c_int, _fp, _fp, _fp, C.c_size_t, _fp]),
        "aline_eig_history_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int, C.c_int]),
        "aline_eig_location_history": (C.c_int, [_fp, _fp, _fp, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                                                  _fp, _fp, _fp, C.c_size_t, _fp]),
        "aline_eig_ces_history": (C.c_int, [_fp, _fp, _fp, C.c_int64, C.c_int, C.c_int, C.c_float, C.c_float, _fp, _fp, _fp, _fp, C.c_size_t, _fp]),
        "aline_head_backward": (C.c_int, [MP, RP, _fp, _fp, _fp, _fp, _fp, GP, _fp, _fp, C.c_size_t, _fp]),
        "aline_encoder_backward": (C.c_int, [MP, RP, _fp, _fp, GP, _fp, _fp, C.c_size_t, _fp]),
        "aline_embed_backward": (C.c_int, [MP, RP, _fp, GP, _fp, C.c_size_t, _fp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    if lib.aline_abi_version() != 5:
        raise RuntimeError("aline_amd: libaline_hip.so ABI version mismatch")
    return lib, sig


lib, SIGNATURES = _load()

# ---- diagnostics: the library never reads the environment; tests and A/B tools set its diagnostic word explicitly -------
DBG = {"DISABLE_FUSED": 1 << 0, "DISABLE_X3": 1 << 2, "DISABLE_S3": 1 << 3,
       "NO_LAYER_TAIL": 1 << 5, "FULL_QKV": 1 << 6, "VALU_ATTENTION": 1 << 7, "S3_GENERIC_EMBED": 1 << 8, "CES_GENERIC": 1 << 9,
       "FUSED_STAMPS": 1 << 10, "NO_BWD_IMAGE_RECOMPUTE": 1 << 11, "NO_BWD_KV_SPARSE": 1 << 12, "SELECT_WORKGROUP": 1 << 13, "S3_SELECT_KERNEL": 1 << 14,
       "NO_BWD_TAIL": 1 << 16, "NO_BWD_ATTN_BLOCK": 1 << 17, "NO_BWD_ACQ": 1 << 18, "NO_BWD_LAYER_FWD": 1 << 19,
       "NO_BWD_LAYER_FWD_FLAT": 1 << 20, "NO_BWD_GMM_FUSED": 1 << 21, "NO_BWD_GMM128": 1 << 22, "NO_BWD_GMM_BATCHED": 1 << 23,
       "NO_BWD_ATTN_MFMA": 1 << 24, "NO_BWD_DW_WALK": 1 << 25, "NO_BWD_GMM_WIDE": 1 << 26, "NO_BWD_SAVED_ACTS": 1 << 27, "BWD_RECOMPUTE_F32": 1 << 28, "BWD_DW_TK2": 1 << 29, "BWD_GRAD_F32": 1 << 30}
DBG_PARAM = {"S3_WAVES": 0, "S3_EPW": 1, "BWD_PREC": 2}


class debug:
    """with _lib.debug("DISABLE_FUSED", S3_WAVES=16): ...   -- sets bits / knobs of the library's diagnostic word for the
    block (names: DBG / DBG_PARAM = the ALINE_DBG_* enumerators of include/aline_hip.h) and restores them afterwards."""

    def __init__(self, *flags, **params):
        self.bits = 0
        for f in flags:
            self.bits |= DBG[f]
        self.params = {DBG_PARAM[k]: int(v) for k, v in params.items()}

    def __enter__(self):
        self.old = lib.aline_debug_get_flags()
        self.old_params = {k: lib.aline_debug_get_param(k) for k in self.params}
        lib.aline_debug_set_flags(self.old | self.bits)
        for k, v in self.params.items():
            check(lib.aline_debug_set_param(k, v), "debug_set_param")
        return self

    def __exit__(self, *exc):
        lib.aline_debug_set_flags(self.old)
        for k, v in self.old_params.items():       # (the values of the enclosing block, not 0: blocks nest)
            lib.aline_debug_set_param(k, v)
        return False


def debug_state():
    """(flags, knobs) of the library's diagnostic word right now."""
    return (int(lib.aline_debug_get_flags()), tuple(int(lib.aline_debug_get_param(k)) for k in sorted(DBG_PARAM.values())))


def debug_env(env):
    """The same from a dict keyed like the former environment switches, e.g. {"ALINE_DISABLE_X3": "1", "ALINE_S3_WAVES": "16",
    "ALINE_BWD_TAIL": "0"} (tests / tools keep their tables of variants in this form)."""
    flags, params = [], {}
    for k, v in env.items():
        name = k[len("ALINE_"):] if k.startswith("ALINE_") else k
        if name in DBG_PARAM:
            params[name] = int(v)
        elif name in DBG:
            if str(v) not in ("", "0"):
                flags.append(name)
        elif "NO_" + name in DBG:               # ALINE_BWD_TAIL = 0  ->  NO_BWD_TAIL
            if str(v) == "0":
                flags.append("NO_" + name)
        else:
            raise KeyError(f"unknown diagnostic switch {k}")
    return debug(*flags, **params)


def _flags_from_environment():
    """Command-line convenience for the tools/ scripts (`ALINE_DBG=DISABLE_FUSED,S3_WAVES=16 python tools/...`), applied once
    at import by THIS module -- the shared library itself has no environment switches."""
    spec = os.environ.get("ALINE_DBG", "")
    bits = 0
    for item in filter(None, (x.strip() for x in spec.split(","))):
        if "=" in item:
            k, v = item.split("=", 1)
            check(lib.aline_debug_set_param(DBG_PARAM[k], int(v)), "debug_set_param")
        else:
            bits |= DBG[item]
    if bits:
        lib.aline_debug_set_flags(bits)




def f16_range_status(ws, device):
    """Reads (host-synchronising) the f16 range status word of a workspace: 0 = clean, bit 0 = an activation operand,
    bit 1 = a weight left f16's range in an F16X3 kernel (include/aline_hip.h)."""
    rc = lib.aline_f16_range_status(ws.data_ptr(), stream_ptr(device))
    if rc < 0:
        check(rc, "f16_range_status")
    return rc


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"aline_amd: {what} failed: {lib.aline_error_string(rc).decode()} ({rc})")


def ptr(t):
    """Device address of a contiguous tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_contiguous(), "aline_amd expects contiguous tensors"
    return t.data_ptr()


def f32(t):
    return None if t is None else t.detach().to(torch.float32).contiguous()


def stream_ptr(device):
    return torch.cuda.current_stream(device).cuda_stream


class Workspace:
    """Grow-only device scratch buffer handed to the C ABI (which never allocates)."""

    def __init__(self):
        self.buf = None

    def get(self, nbytes, device):
        if self.buf is None or self.buf.numel() < nbytes or self.buf.device != device:
            self.buf = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=device)
        return self.buf


_flags_from_environment()

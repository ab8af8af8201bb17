"""CPU: the C-ABI library loads and exports every symbol include/aline_hip.h declares; the ctypes
struct mirrors have the C layout.  No compute calls (no GPU here)."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "aline_hip.h")
LIB = os.path.join(ROOT, "aline_amd", "csrc", "libaline_hip.so")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(aline_[a-z_0-9]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(LIB):
        sys.path.insert(0, ROOT)
        import __graft_entry__ as g
        g.build()
    return ctypes.CDLL(LIB)


def test_exports_every_declared_symbol(lib):
    names = declared_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/aline_hip.h but not exported"


def test_abi_version_and_errors(lib):
    lib.aline_abi_version.restype = ctypes.c_int
    assert lib.aline_abi_version() == 5
    lib.aline_error_string.restype = ctypes.c_char_p
    assert lib.aline_error_string(0) == b"ok"
    assert b"workspace" in lib.aline_error_string(-3)


def test_only_the_c_abi_is_exported_and_the_library_ignores_the_environment():
    """-fvisibility=hidden: the dynamic symbol table holds the functions of include/aline_hip.h and nothing else of the
    library's code (the C++ device stubs of round 2 are gone; what remains besides them are hipcc's kernel HANDLE objects,
    data symbols the HIP runtime registers kernels by).  And the library has no environment switches: `getenv` is not among
    its imports (diagnostics go through aline_debug_set_flags)."""
    out = subprocess.check_output(["nm", "-D", LIB]).decode().splitlines()
    funcs = [l.split()[-1] for l in out if len(l.split()) == 3 and l.split()[1] in "Tt"]
    assert sorted(funcs) == declared_functions(), sorted(set(funcs) ^ set(declared_functions()))
    undefined = [l.split()[-1].split("@")[0] for l in out if len(l.split()) == 2 and l.split()[0] == "U"]
    assert "getenv" not in undefined and "secure_getenv" not in undefined


def test_debug_word_and_kernel_name(lib):
    """The diagnostic word (0 by default) steers the path selection; aline_rollout_kernel_name reports the dominant kernel
    with the template arguments of the launch shape -- host logic only."""
    from aline_amd import _lib
    L = _lib.lib
    assert L.aline_debug_get_flags() == 0
    m = _small_model()
    m.precision = _lib.PREC["f16x3"]
    r = _lib.AlineRollout()
    r.B, r.P, r.n_ctx0, r.n_target_data, r.T = 1000, 201, 1, 0, 30
    buf = ctypes.create_string_buffer(128)
    assert L.aline_rollout_kernel_name(ctypes.byref(m), ctypes.byref(r), buf, 128) == 4
    assert buf.value == b"s3::step_kernel<128, 12, 2, false, false>"
    with _lib.debug("DISABLE_S3"):
        assert L.aline_rollout_path(ctypes.byref(m), ctypes.byref(r)) == 0
    with _lib.debug(S3_WAVES=16):
        L.aline_rollout_kernel_name(ctypes.byref(m), ctypes.byref(r), buf, 128)
        assert buf.value == b"s3::step_kernel<128, 16, 2, false, false>"
    with _lib.debug_env({"ALINE_DISABLE_S3": "1", "ALINE_BWD_TAIL": "0"}):
        assert L.aline_debug_get_flags() == _lib.DBG["DISABLE_S3"] | _lib.DBG["NO_BWD_TAIL"]
    assert L.aline_debug_get_flags() == 0
    assert L.aline_rollout_path(ctypes.byref(m), ctypes.byref(r)) == 4
    r.T, r.n_target_data = 50, 100                       # cfg3: up to 153 keys -> the 8-wave variant
    m.n_theta, m.embedding_type = 3, 2
    r.B = 512
    L.aline_rollout_kernel_name(ctypes.byref(m), ctypes.byref(r), buf, 128)
    assert buf.value == b"s3::step_kernel<128, 8, 5, false, false>"
    assert L.aline_debug_set_param(99, 1) == -1
    assert L.aline_f16_range_offset() == 0
    # every enumerator of the header's ALINE_DBG_* list has the value the Python mirror uses
    hdr = open(HEADER).read()
    for name, bit in _lib.DBG.items():
        mo = re.search(rf"ALINE_DBG_{name} = 1u << (\d+)", hdr)
        assert mo and (1 << int(mo.group(1))) == bit, name


def test_aline_dbg_environment_convenience_of_the_python_side():
    """`ALINE_DBG=NAME,KEY=value` (tools/ scripts) is applied by aline_amd._lib at import -- flags and parameters."""
    code = ("import aline_amd._lib as L; print(L.lib.aline_debug_get_flags() & L.DBG['DISABLE_FUSED'] != 0)")
    env = dict(os.environ, ALINE_DBG="DISABLE_FUSED,S3_WAVES=12,S3_EPW=3")
    out = subprocess.run([sys.executable, "-c", code], env=env, cwd=ROOT, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip() == "True"
    bad = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, ALINE_DBG="NO_SUCH_SWITCH"), cwd=ROOT,
                         capture_output=True, text=True)
    assert bad.returncode != 0


def test_struct_layout_matches_c(tmp_path):
    """sizeof/offsetof from a tiny C program must equal the ctypes mirrors."""
    from aline_amd import _lib
    prog = tmp_path / "layout.c"
    prog.write_text(f'''
#include <stdio.h>
#include <stddef.h>
#include "{HEADER}"
int main(void) {{
  printf("%zu %zu %zu\\n", sizeof(aline_model), sizeof(aline_step), sizeof(aline_rollout));
  printf("%zu %zu %zu %zu\\n", offsetof(aline_model, x_w1), offsetof(aline_model, in_proj_w),
         offsetof(aline_model, acq_w1), offsetof(aline_model, gmm_b2));
  printf("%zu %zu %zu\\n", offsetof(aline_step, select_mode), offsetof(aline_step, idx),
         offsetof(aline_step, encoding));
  printf("%zu %zu %zu %zu\\n", offsetof(aline_rollout, select_mode), offsetof(aline_rollout, time_token_T),
         offsetof(aline_rollout, ev_kernel_stop), offsetof(aline_rollout, ev_kernel_step));
  return 0;
}}''')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-o", str(exe), str(prog)])
    out = subprocess.check_output([str(exe)]).decode().split()
    got = [int(v) for v in out]
    M, S, R = _lib.AlineModel, _lib.AlineStep, _lib.AlineRollout
    exp = [ctypes.sizeof(M), ctypes.sizeof(S), ctypes.sizeof(R),
           M.x_w1.offset, M.in_proj_w.offset, M.acq_w1.offset, M.gmm_b2.offset,
           S.select_mode.offset, S.idx.offset, S.encoding.offset,
           R.select_mode.offset, R.time_token_T.offset, R.ev_kernel_stop.offset, R.ev_kernel_step.offset]
    assert got == exp


def test_workspace_query_is_pure_host(lib):
    """*_workspace_bytes touches no device state: usable without a GPU."""
    from aline_amd import _lib
    m = _lib.AlineModel()
    m.dim_x, m.dim_y, m.d, m.F, m.H, m.L, m.C = 2, 1, 32, 128, 4, 3, 10
    m.n_theta, m.embedding_type = 2, 1
    s = _lib.AlineStep()
    s.B, s.n_ctx, s.n_query, s.n_target_data = 1000, 1, 200, 0
    n = _lib.lib.aline_step_workspace_bytes(ctypes.byref(m), ctypes.byref(s))
    assert n > 1000 * 203 * 32 * 4
    m.d = 33                                     # unsupported width -> 0
    assert _lib.lib.aline_step_workspace_bytes(ctypes.byref(m), ctypes.byref(s)) == 0


def test_product_does_not_import_oracle():
    """The product package must never route through the oracle (parity would be void)."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "aline_amd")):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dirpath, f)).read()
                assert "aline_oracle" not in txt and "import oracle" not in txt, os.path.join(dirpath, f)


def _small_model():
    from aline_amd import _lib
    m = _lib.AlineModel()
    m.dim_x, m.dim_y, m.d, m.F, m.H, m.L, m.C = 2, 1, 32, 128, 4, 3, 10
    m.n_theta, m.embedding_type = 2, 1
    fake = 0x1000                                    # never dereferenced: validation fails first
    for n in ("x_w1", "x_b1", "x_w2", "x_b2", "y_w1", "y_b1", "y_w2", "y_b2", "theta_tokens", "acq_w1", "acq_b1",
              "acq_w2", "acq_b2"):
        setattr(m, n, fake)
    for l in range(3):
        for n in ("in_proj_w", "in_proj_b", "out_proj_w", "out_proj_b", "lin1_w", "lin1_b", "lin2_w", "lin2_b",
                  "norm1_w", "norm1_b", "norm2_w", "norm2_b"):
            getattr(m, n)[l] = fake
    for c in range(10):
        for n in ("gmm_w1", "gmm_b1", "gmm_w2", "gmm_b2"):
            getattr(m, n)[c] = fake
    return m


def test_error_codes_before_any_launch(lib):
    """Argument / shape / workspace validation happens on the host before the first kernel launch, so
    the error behaviour is testable without a GPU: bad calls return ALINE_E* and never throw."""
    from aline_amd import _lib
    L = _lib.lib
    m = _small_model()
    s = _lib.AlineStep()
    s.B, s.n_ctx, s.n_query, s.n_target_data = 4, 1, 20, 0
    s.context_x = s.context_y = s.query_x = s.target_all = 0x1000
    ws = 0x2000
    byref = ctypes.byref
    assert L.aline_step_forward(byref(m), byref(s), ws, 16, None) == -3            # ALINE_EWORKSPACE
    assert L.aline_step_forward(None, byref(s), ws, 1 << 30, None) == -1           # ALINE_EINVAL
    s.n_ctx = 0                                                                     # n_ctx >= 1 (edge cases)
    assert L.aline_step_forward(byref(m), byref(s), ws, 1 << 30, None) == -1
    s.n_ctx = 1
    s.select_mode = 1                                                               # SAMPLE without uniforms
    assert L.aline_step_forward(byref(m), byref(s), ws, 1 << 30, None) == -1
    s.select_mode = 0
    m.d = 48                                                                        # d % 32 != 0
    assert L.aline_step_forward(byref(m), byref(s), ws, 1 << 30, None) == -2       # ALINE_EUNSUPPORTED
    m.d = 32
    m.dim_y = 2                                                                     # GMM head is single-output
    assert L.aline_step_forward(byref(m), byref(s), ws, 1 << 30, None) == -2
    m.dim_y = 1
    m.lin1_w[1] = None                                                              # missing weight pointer
    assert L.aline_step_forward(byref(m), byref(s), ws, 1 << 30, None) == -1
    r = _lib.AlineRollout()
    r.B, r.P, r.n_ctx0, r.T = 4, 21, 1, 30                                          # n_ctx0 + T > P
    r.point_x = r.point_y = r.role = 0x1000
    m = _small_model()
    assert L.aline_rollout_forward(byref(m), byref(r), ws, 1 << 40, None) == -1
    assert L.aline_eig_finalize(0x1000, 1, 4, None, None, ws, 1 << 20, None) == -1  # needs L >= 1
    assert L.aline_cholesky_upper(None, 4, 1, None, None) == -1


def test_rollout_path_selection_is_pure_host(lib):
    """aline_rollout_path: which implementation aline_rollout_forward takes -- host logic only, so the dispatch rules
     are testable without a GPU."""
    from aline_amd import _lib
    L = _lib.lib
    byref = ctypes.byref
    GENERIC, FUSED, X3, S3 = 0, 1, 3, 4

    def rollout(B=1000, P=201, n_ctx0=1, n_td=0, T=30):
        r = _lib.AlineRollout()
        r.B, r.P, r.n_ctx0, r.n_target_data, r.T = B, P, n_ctx0, n_td, T
        return r

    m = _small_model()                                      # d = 32, F = 128, 4 heads, theta mode
    m.precision = _lib.PREC["f32"]
    assert L.aline_rollout_path(byref(m), byref(rollout())) == FUSED          # round 1's exact-fp32 kernel
    m.precision = _lib.PREC["f16x3"]
    assert L.aline_rollout_path(byref(m), byref(rollout())) == S3             # the benchmarked path
    assert L.aline_rollout_path(byref(m), byref(rollout(T=160))) == GENERIC   # 1 + 159 + 2 keys > 160
    assert L.aline_rollout_path(byref(m), byref(rollout(B=2, P=2001, T=35))) == S3   # the evaluation protocol's n_query = 2000 (README.md:45)
    m.n_theta, m.embedding_type = 3, 2                                       # mix mode (cfg3): 1 + 49 + 103 keys
    assert L.aline_rollout_path(byref(m), byref(rollout(B=512, n_td=100, T=50))) == S3
    m.precision = _lib.PREC["f32"]
    assert L.aline_rollout_path(byref(m), byref(rollout(B=512, n_td=100, T=50))) == GENERIC    # fused kernel: theta mode only
    m.precision = _lib.PREC["f16x3"]
    m.time_token = 1                                                          # a time token is a per-step bias of the head: same path
    assert L.aline_rollout_path(byref(m), byref(rollout(B=512, n_td=100, T=50))) == S3
    m.time_token = 0
    m.H = 8                                                                   # head_dim 4: not an s3 shape
    assert L.aline_rollout_path(byref(m), byref(rollout(B=512, n_td=100, T=50))) == GENERIC
    w = _small_model()
    w.d, w.F, w.H = 256, 1024, 8
    w.precision = _lib.PREC["f16x3"]
    assert L.aline_rollout_path(byref(w), byref(rollout())) == X3
    w.precision = _lib.PREC["bf16"]
    assert L.aline_rollout_path(byref(w), byref(rollout())) == GENERIC           # (ABI 5: the bf16 `wide` kernels are gone)
    assert L.aline_rollout_path(byref(w), byref(rollout(T=70))) == GENERIC    # 72 keys > 64
    w.d, w.F, w.precision = 512, 128, _lib.PREC["f16x3"]                      # the psychometric configuration's width (cfg5)
    assert L.aline_rollout_path(byref(w), byref(rollout(B=256))) == 5         # ALINE_PATH_X5
    w.F = 2048
    assert L.aline_rollout_path(byref(w), byref(rollout(B=256))) == 5
    assert L.aline_rollout_path(byref(w), byref(rollout(B=256, T=70))) == GENERIC
    buf = ctypes.create_string_buffer(64)
    assert L.aline_rollout_kernel_name(byref(w), byref(rollout(B=256)), buf, 64) == 5 and buf.value == b"x5::layer_kernel<true>"
    w.d = 48
    assert L.aline_rollout_path(byref(w), byref(rollout())) == -2             # ALINE_EUNSUPPORTED, as the forward would say
    assert L.aline_rollout_path(None, byref(rollout())) == -1
    # aline_rollout.saved_acts: sized for the s3 path of the fused-backward width only ([2 L + 1][T B N d] floats), 0 elsewhere
    m = _small_model()
    m.precision = _lib.PREC["f16x3"]
    r = rollout(B=10, P=21, T=5)
    assert L.aline_rollout_saved_acts_bytes(byref(m), byref(r)) == (2 * m.L + 1) * 5 * 10 * (21 + m.n_theta) * 32 * 4
    m.precision = _lib.PREC["f32"]
    assert L.aline_rollout_saved_acts_bytes(byref(m), byref(r)) == 0          # the exact-fp32 fused rollout does not keep them
    assert L.aline_rollout_saved_acts_bytes(byref(w), byref(r)) == 0 and L.aline_rollout_saved_acts_bytes(None, byref(r)) == 0

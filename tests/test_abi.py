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
    assert lib.aline_abi_version() == 1
    lib.aline_error_string.restype = ctypes.c_char_p
    assert lib.aline_error_string(0) == b"ok"
    assert b"workspace" in lib.aline_error_string(-3)


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
  printf("%zu %zu %zu\\n", offsetof(aline_rollout, select_mode), offsetof(aline_rollout, time_token_T),
         offsetof(aline_rollout, ev_kernel_stop));
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
           R.select_mode.offset, R.time_token_T.offset, R.ev_kernel_stop.offset]
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

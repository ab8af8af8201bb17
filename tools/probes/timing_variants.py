#!/usr/bin/env python3
"""Timing experiments as PROBE builds: the shipped headers carry no experiment switches (hygiene).  A variant is a
list of text edits applied to a scratch copy of aline_amd/csrc/, compiled into aline_amd/csrc/variants/lib_<name>.so and run with
ALINE_HIP_LIB=<that file> (results of a variant with parts removed are garbage: only the time is read).

    python tools/probes/timing_variants.py list
    python tools/probes/timing_variants.py build x3_no_dma x3_no_barrier ...      (cross-compiles here; the .so travels with gpurun)
    tools/x3_variants.sh run x3_no_dma ...                                         (on the GPU box)

Every edit must match exactly once, so a variant that no longer applies to the current source fails loudly."""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "aline_amd", "csrc")

# name -> [(file, old text, new text), ...]
VARIANTS = {
    # the weight stream of the x3 / x5 layer kernels without its LDS-DMA instructions (r03: 661 -> 590 us)
    "x3_no_dma": [("x3_impl.h", "        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void *)(d + i * 1024), 16, lane_off, wave_off + i * 1024, 0, 0);\n",
                   "        asm volatile(\"\" :: \"v\"(d + i * 1024));\n")],
    # ... without the landing wait / without the chunk barrier (r03: 675 us without the barrier)
    "x3_no_wait": [("x3_impl.h", "    wait_vmcnt<PIECES_PER_WAVE *(PD - 2)>();\n    X3_LAP(*this, 3);\n", "    X3_LAP(*this, 3);\n")],
    "x3_no_barrier": [("x3_impl.h", "    X3_LAP(*this, 3);\n    __builtin_amdgcn_s_barrier();\n", "    X3_LAP(*this, 3);\n")],
    # the generic F16X3 GEMM: no operand split / no prefetch of the next k-step
    "gemm_no_split": [("gemm.h", "    split2_f16(v.x * scale, v.y * scale, h0, l0);\n    split2_f16(v.z * scale, v.w * scale, h1, l1);\n",
                       "    h0 = __float_as_uint(v.x); l0 = __float_as_uint(v.y); h1 = __float_as_uint(v.z); l1 = __float_as_uint(v.w);\n")],
    "gemm_no_gload": [("gemm.h", "    if (k0 + GEMM_BK < a.K) load_step(k0 + GEMM_BK);\n", "")],
    # s3 step kernel: the next tile's rows requested a tile ahead (r03: 2 % slower)
    "s3_prefetch": [("aline_hip.hip", "  constexpr bool PF = false;\n", "  constexpr bool PF = NW >= 12;\n")],
}


def build(name):
    edits = VARIANTS[name]
    out = os.path.join(CSRC, "variants")
    os.makedirs(out, exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, "csrc")
        shutil.copytree(CSRC, src, ignore=shutil.ignore_patterns("*.so", "variants", "build.log"))
        os.makedirs(os.path.join(tmp, "include"), exist_ok=True)
        for f, old, new in edits:
            p = os.path.join(src, f)
            s = open(p).read()
            if s.count(old) != 1:
                raise SystemExit(f"variant {name}: edit of {f} matches {s.count(old)} times (expected 1): {old[:60]!r}")
            open(p, "w").write(s.replace(old, new))
        # the sources include "../../include/aline_hip.h" relative to csrc: mirror that layout
        os.makedirs(os.path.join(tmp, "a", "b"), exist_ok=True)
        shutil.move(src, os.path.join(tmp, "a", "b", "csrc"))
        shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tmp, "include"), dirs_exist_ok=True)
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-fno-slp-vectorize", "-Wno-unused-function",
               "-shared", "-o", os.path.join(out, f"lib_{name}.so"), "aline_hip.hip"]
        subprocess.check_call(cmd, cwd=os.path.join(tmp, "a", "b", "csrc"))
    print("built", os.path.join(out, f"lib_{name}.so"))


if __name__ == "__main__":
    if len(sys.argv) < 2 or sys.argv[1] == "list":
        for k, v in VARIANTS.items():
            print(k, "->", ", ".join(sorted({f for f, _, _ in v})))
    elif sys.argv[1] == "build":
        for n in sys.argv[2:]:
            build(n)

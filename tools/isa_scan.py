#!/usr/bin/env python3
"""Scheduling audit of the gfx950 code object: per kernel and basic block, the order of global loads (L), LDS-DMA pieces (D), counted / full
vmcnt waits (w / W), MFMAs (M), workgroup barriers (B) and scratch accesses (S) as run-length strings.

    python tools/isa_scan.py                     # the 40 blocks with the most "load runs issued behind a wait"
    python tools/isa_scan.py gemm_tn_block       # every MFMA / load block of the kernels whose demangled name contains the argument

What it is for (profiles/r03_x3_timing_experiments.txt): hipcc keeps live ranges short by requesting a group of independent loads only when the
arithmetic on the group before has its operands -- `L12 w M32 L12 w M32 ...` inside one block means that many DEPENDENT round trips where the source
meant one (`gemm_tn_block_kernel`: four per 64 rows; pinning the loads in front of the MFMAs with `__builtin_amdgcn_sched_barrier(0)` was worth 13 %,
rotating the loop so that they are requested a phase ahead another 5 %).  A block of the form `L1 w M2 L1 w M1 ...` is a just-in-time stream the
compiler built from a prefetch that did not fit the register file (the K / V pairs in `x5::layer_kernel`'s attention)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def disassemble():
    src = os.path.join(ROOT, "aline_amd", "csrc", "aline_hip.hip")
    return subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-slp-vectorize", "-S",
                           "--cuda-device-only", "-o", "-", src], capture_output=True, text=True, check=True).stdout.splitlines()


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return dict(zip(names, out))


def blocks(lines):
    kern, blk, cur = None, None, None
    for l in lines:
        m = re.match(r"^(_Z\w+):", l)
        if m:
            kern, blk, cur = m.group(1), "entry", []
            yield_key = (kern, blk)
            res.append((yield_key, cur))
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m and kern:
            blk, cur = m.group(1), []
            res.append(((kern, blk), cur))
            continue
        if kern is None:
            continue
        s = l.strip()
        if s.startswith("s_endpgm"):
            kern = None
        elif re.match(r"(global_load|buffer_load)_dword", s):
            cur.append("D" if " lds" in s else "L")
        elif s.startswith("scratch_"):
            cur.append("S")
        elif s.startswith("s_waitcnt") and "vmcnt" in s:
            cur.append("W" if re.search(r"vmcnt\(0\)", s) else "w")
        elif s.startswith("v_mfma"):
            cur.append("M")
        elif s.startswith("s_barrier"):
            cur.append("B")


def runs(seq):
    out = []
    for x in seq:
        if out and out[-1][0] == x:
            out[-1][1] += 1
        else:
            out.append([x, 1])
    return out


if __name__ == "__main__":
    needle = sys.argv[1] if len(sys.argv) > 1 else None
    res = []
    blocks(disassemble())
    names = demangle(sorted({k for (k, _), _ in res}))
    rows = []
    for (k, b), seq in res:
        if seq.count("L") + seq.count("M") == 0:
            continue
        r = runs(seq)
        chained, seen_wait = 0, False
        for a, _ in r:
            if a in "wW":
                seen_wait = True
            elif a == "L":
                chained += seen_wait
                seen_wait = False
        rows.append((chained, seq.count("L"), seq.count("M"), names.get(k, k), b, " ".join(f"{a}{n if n > 1 else ''}" for a, n in r)))
    if needle:
        for c, nl, nm, k, b, txt in rows:
            if needle in k and (nm >= 16 or nl >= 8):
                print(f"{k[:90]} {b}: loads {nl}, MFMAs {nm}, load runs behind a wait {c}\n    {txt[:1500]}")
    else:
        rows.sort(reverse=True)
        for c, nl, nm, k, b, _ in rows[:40]:
            print(f"{c:4d} load runs behind a wait, {nl:4d} loads, {nm:5d} MFMAs  {k[:100]} {b}")

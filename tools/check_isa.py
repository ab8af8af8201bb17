#!/usr/bin/env python3
"""Build-time audit of the gfx950 code object (python tools/check_isa.py; exits non-zero on a finding).

1. Hand-managed MFMA -> VALU hazards.  wide_step_kernel keeps its accumulators in AGPRs through opaque inline-asm
   `v_accvgpr_read_b32` (wide_step.h: acc_rd), which the compiler's hazard recogniser cannot see; the source guards every
   such read with mfma_drain().  MEASURED requirement (tools/probes/mfma_valu_hazard.hip on MI355X): 7 wait states between a
   v_mfma_f32_16x16x32_* and ANY VALU reader of its result -- v_max_i32, v_max_f32 and v_accvgpr_read_b32 alike -- and hipcc
   itself pads 8.  This script walks the disassembly of every kernel and fails if an inline-asm v_accvgpr_read_b32 (between
   ;;#ASMSTART / ;;#ASMEND) can be reached from a v_mfma with fewer than REQUIRED wait states in between (straight-line
   distance inside a basic block; a block boundary counts as unknown = 0 states, so a read at the top of a block needs its
   pad in the same block).
2. No scalar-memory instruction and no scratch access inside the x3 pipelines' hot loops is NOT required any more (the
   x3 reads are compiler-visible), but spills in the x3 layer kernel are reported.
"""
import re
import subprocess
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = 8          # measured 7, one state of margin


def wait_states(line):
    m = re.match(r"\s*s_nop\s+(\d+)", line)
    if m:
        return int(m.group(1)) + 1
    return 1


def main():
    src = os.path.join(ROOT, "aline_amd", "csrc", "aline_hip.hip")
    asm = subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-slp-vectorize", "-S", "--cuda-device-only",
                          "-o", "-", src], capture_output=True, text=True, check=True).stdout
    lines = asm.splitlines()
    INF = 10 ** 6
    label_in = {}                 # label -> fewest wait states since a v_mfma over the branches that jump to it
    findings, spills, n_reads = [], {}, 0
    for sweep in range(4):        # fixed point over backward branches (loops)
        findings, spills, n_reads = [], {}, 0
        kernel, since, in_asm = None, INF, False
        for ln, line in enumerate(lines, 1):
            s = line.strip()
            m = re.match(r"^(_Z\w+):", line)
            if m:
                kernel, since = m.group(1), INF
                continue
            m = re.match(r"^(\.LBB\w+):", line)
            if m:                 # a label: the worst case over the fall-through path and every branch into it
                since = min(since, label_in.get((kernel, m.group(1)), INF))
                continue
            if not s or s.startswith(".") or (s.startswith(";") and "ASM" not in s):
                continue
            if ";;#ASMSTART" in s:
                in_asm = True
                continue
            if ";;#ASMEND" in s:
                in_asm = False
                continue
            if "scratch_" in s and kernel and "x312layer_kernel" in kernel:
                spills[kernel] = spills.get(kernel, 0) + 1
            if s.startswith("v_mfma"):
                since = 0
                continue
            if in_asm and s.startswith("v_accvgpr_read_b32"):
                n_reads += 1
                if since < REQUIRED:
                    findings.append(f"{kernel}: line {ln}: v_accvgpr_read_b32 {since} wait states after a v_mfma (need {REQUIRED})")
            m = re.match(r"s_(?:cbranch_\w+|branch)\s+(\.LBB\w+)", s)
            if m:
                key = (kernel, m.group(1))
                label_in[key] = min(label_in.get(key, INF), since + 1)
                if s.startswith("s_branch"):
                    since = INF       # nothing falls through an unconditional branch
                    continue
            since = min(INF, since + wait_states(s))
    print(f"check_isa: {n_reads} inline-asm v_accvgpr_read_b32 audited, required distance to the last v_mfma: {REQUIRED} wait states")
    for k, n in spills.items():
        print(f"check_isa: note: {n} scratch instructions in {k}")
    for f in findings:
        print("check_isa: FINDING:", f)
    return 1 if findings else 0


if __name__ == "__main__":
    sys.exit(main())

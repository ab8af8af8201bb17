#!/usr/bin/env python3
"""Build-time audit of the gfx950 code object (python tools/check_isa.py; exits non-zero on a finding).

1. Hand-managed MFMA -> VALU hazards.  A kernel that keeps accumulators in AGPRs through opaque inline-asm
   `v_accvgpr_read_b32` (the bf16 step kernel of rounds 1-3 did; none does since round 4, the audit stays for the next one)
   hides those reads from the compiler's hazard recogniser and must pad them itself.  MEASURED requirement (tools/probes/mfma_valu_hazard.hip on MI355X): 7 wait states between a
   v_mfma_f32_16x16x32_* and ANY VALU reader of its result -- v_max_i32, v_max_f32 and v_accvgpr_read_b32 alike -- and hipcc
   itself pads 8.  This script walks the disassembly of every kernel and fails if an inline-asm v_accvgpr_read_b32 (between
   ;;#ASMSTART / ;;#ASMEND) can be reached from a v_mfma with fewer than REQUIRED wait states in between (straight-line
   distance inside a basic block; a block boundary counts as unknown = 0 states, so a read at the top of a block needs its
   pad in the same block).
2. No scalar-memory instruction and no scratch access inside the x3 pipelines' hot loops is NOT required any more (the
   x3 reads are compiler-visible), but spills in the x3 layer kernel are reported.
3. The code shape behind round 1's run-to-run irreproducibility: an integer max on float bits (`v_max_i32` /
   `v_max_u32`: the one-instruction ReLU) whose result is a source of a packed fp32 instruction (`v_pk_fma_f32`, `v_pk_mul_f32`,
   `v_pk_add_f32`) in the same basic block.  ReLU is fmaxf everywhere and the library is built with -fno-slp-vectorize, so the
   pattern must not occur: any occurrence fails the build.  (Packed fp32 instructions as such remain -- they come from f32x4
   source arithmetic -- and are counted per kernel as a note.)
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
        imax_regs, n_pk = set(), 0   # VGPRs last written by an integer max in the current basic block; packed-fp32 instructions seen
        for ln, line in enumerate(lines, 1):
            s = line.strip()
            m = re.match(r"^(_Z\w+):", line)
            if m:
                kernel, since = m.group(1), INF
                continue
            m = re.match(r"^(\.LBB\w+):", line)
            if m:                 # a label: the worst case over the fall-through path and every branch into it
                since = min(since, label_in.get((kernel, m.group(1)), INF))
                imax_regs = set()
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
            # (3) integer max feeding a packed fp32 instruction
            mo = re.match(r"(v_\w+)\s+(.*)", s)
            if mo:
                ops = [o.strip() for o in mo.group(2).split(",")]
                regs = lambda o: ({int(o[1:])} if re.fullmatch(r"v\d+", o) else                       # noqa: E731
                                  set(range(int(o[2:-1].split(":")[0]), int(o[2:-1].split(":")[1]) + 1)) if re.fullmatch(r"v\[\d+:\d+\]", o) else set())
                dst = regs(ops[0]) if ops else set()
                if mo.group(1).startswith("v_pk_") and mo.group(1).endswith("_f32"):
                    n_pk += 1
                    used = set().union(*[regs(o) for o in ops[1:]]) if len(ops) > 1 else set()
                    if used & imax_regs:
                        findings.append(f"{kernel}: line {ln}: {mo.group(1)} reads v{sorted(used & imax_regs)} written by an integer max (the round-1 irreproducibility shape)")
                if mo.group(1).startswith(("v_max_i32", "v_max_u32")):
                    imax_regs |= dst
                else:
                    imax_regs -= dst
            ml = re.match(r"(?:global_load|ds_read|buffer_load|scratch_load|flat_load)\w*\s+(v\d+|v\[\d+:\d+\])", s)
            if ml:                # a load overwrites its destination registers
                o = ml.group(1)
                imax_regs -= ({int(o[1:])} if ":" not in o else set(range(int(o[2:-1].split(":")[0]), int(o[2:-1].split(":")[1]) + 1)))
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
    print(f"check_isa: {n_pk} packed-fp32 instructions in the code object, none fed by an integer max" if not any("integer max" in f for f in findings)
          else "check_isa: packed-fp32 instruction fed by an integer max FOUND")
    for k, n in spills.items():
        print(f"check_isa: note: {n} scratch instructions in {k}")
    for f in findings:
        print("check_isa: FINDING:", f)
    return 1 if findings else 0


if __name__ == "__main__":
    sys.exit(main())

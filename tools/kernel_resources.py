#!/usr/bin/env python3
"""Per-kernel register / spill / scratch table out of hipcc's -Rpass-analysis=kernel-resource-usage remarks
(the Makefile saves them next to the library: aline_amd/csrc/build.log -> kernel_resources.txt).
    python tools/kernel_resources.py aline_amd/csrc/build.log [substring filter]"""
import re
import subprocess
import sys


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout
        return out.splitlines()
    except Exception:
        return names


def parse(path):
    rows, cur = [], None
    for line in open(path, errors="replace"):
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+) \[-Rpass", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = m.group(2)
    return rows


def main():
    rows = parse(sys.argv[1])
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    names = demangle([r["name"] for r in rows])
    print(f"{'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'vspill':>6} {'sspill':>6} {'scratch':>7} {'occ':>3}  kernel")
    for r, n in sorted(zip(rows, names), key=lambda x: x[1]):
        if flt and flt not in n:
            continue
        print(f"{r.get('VGPRs', '?'):>5} {r.get('AGPRs', '?'):>5} {r.get('TotalSGPRs', '?'):>5} {r.get('VGPRs Spill', '?'):>6} "
              f"{r.get('SGPRs Spill', '?'):>6} {r.get('ScratchSize', '?'):>7} {r.get('Occupancy', '?'):>3}  {n[:150]}")


if __name__ == "__main__":
    main()

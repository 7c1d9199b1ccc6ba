#!/usr/bin/env python3
"""Registers / scratch of every kernel in a device assembly file kept by build.py (csrc/build/*-gfx950.s).
usage: tools/kernel_regs.py gemm_stream [substring]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    path = os.path.join(ROOT, "wfl-asr_amd", "csrc", "build", sys.argv[1] + "-hip-amdgcn-amd-amdhsa-gfx950.s")
    s = open(path, errors="replace").read()
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    rows = []
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, re.S):
        name, body = m.group(1), m.group(2)
        v = re.search(r"\.amdhsa_next_free_vgpr (\d+)", body).group(1)
        sp = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body).group(1)
        rows.append((name, v, sp))
    names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
    for (name, v, sp), dn in zip(rows, names):
        if filt in dn:
            print("vgpr %3s  scratch %4s  %s" % (v, sp, dn[:150]))


if __name__ == "__main__":
    main()

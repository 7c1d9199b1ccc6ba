#!/usr/bin/env python3
"""Where a kernel's register spills are reloaded, and whether the reload is followed by an `s_waitcnt vmcnt` (a wait behind everything the
wave has in flight: operand DMA, residual loads, epilogue stores).  usage: tools/asm_spills.py gemm_stream '<0, 6, true, 0, true'"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    path = os.path.join(ROOT, "wfl-asr_amd", "csrc", "build", sys.argv[1] + "-hip-amdgcn-amd-amdhsa-gfx950.s")
    s = open(path, errors="replace").read()
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    starts = [(m.start(), m.group(1)) for m in re.finditer(r"^(_Z\S+):\s+; @", s, re.M)]
    names = subprocess.run(["c++filt"], input="\n".join(n for _, n in starts), capture_output=True, text=True).stdout.split("\n")
    for (st, name), dn in zip(starts, names):
        if filt not in dn:
            continue
        body = s[st:s.find(".Lfunc_end", st)]
        lines = [l.strip() for l in body.split("\n")]
        mf = [i for i, l in enumerate(lines) if l.startswith("v_mfma")]
        print(dn[:110], "| lines", len(lines), "| first / last MFMA at", mf[0] if mf else None, mf[-1] if mf else None)
        for i, l in enumerate(lines):
            if "scratch_load" in l:
                w = next((x for x in lines[i + 1:i + 12] if x.startswith("s_waitcnt") and "vmcnt" in x), "")
                print("   reload at %5d  %-48s %s" % (i, l.split(";")[0][:48], w))


if __name__ == "__main__":
    main()

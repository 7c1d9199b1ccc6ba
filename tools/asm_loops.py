#!/usr/bin/env python3
"""Per kernel of a device assembly file kept by build.py: the basic blocks that hold MFMAs, with their instruction mix (spills,
LDS reads, LDS-DMA, waits, barriers).  usage: tools/asm_loops.py gemm_mx [kernel-substring]"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    path = os.path.join(ROOT, "wfl-asr_amd", "csrc", "build", sys.argv[1] + "-hip-amdgcn-amd-amdhsa-gfx950.s")
    s = open(path, errors="replace").read()
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    starts = [(m.start(), m.group(1)) for m in re.finditer(r"^(_Z\S+):\s+; @", s, re.M)]
    for st, name in starts:
        if filt not in name:
            continue
        en = s.find(".Lfunc_end", st)
        body = s[st:en]
        blocks = re.split(r"\n(?=\.LBB\d+_\d+:)", body)
        print(name, "| scratch ops", body.count("scratch_"), "| lines", body.count("\n"))
        for b in blocks:
            nm = len(re.findall(r"\bv_mfma", b))
            if nm:
                print("   %-10s mfma %3d  scratch %3d  ds_read %3d  lds-dma %2d  waitcnt %3d  barrier %d  valu %4d  lines %4d" % (
                    b.split(":")[0][:10], nm, b.count("scratch_"), len(re.findall(r"\bds_read", b)), b.count("global_load_lds"),
                    b.count("s_waitcnt"), b.count("s_barrier"), len(re.findall(r"^\s+v_(?!mfma)", b, re.M)), b.count("\n")))


if __name__ == "__main__":
    main()

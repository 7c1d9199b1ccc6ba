#!/usr/bin/env python3
"""Diagnostic (not product): a second build of libwfl_asr_hip.so with ONE translation unit recompiled under extra -D flags, for same-box
A/B runs of the whole model (WFL_LIB_PATH=tools/_diag/libwfl_<name>.so python bench.py ...).
usage: ab_lib.py <name> <file.hip> [-DFLAG ...]"""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("wfl_build", os.path.join(ROOT, "wfl-asr_amd", "build.py"))
B = importlib.util.module_from_spec(spec)
spec.loader.exec_module(B)


def main():
    name, unit, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
    B.build_library()
    out = os.path.join(ROOT, "tools", "_diag")
    os.makedirs(out, exist_ok=True)
    obj = os.path.join(out, f"ab_{name}_{unit[:-4]}.o")
    hipcc = B._hipcc()
    base = [f for f in B.FLAGS if f != "-save-temps" and not f.startswith("-save-temps")]
    cmd = [hipcc, *base, *B.PER_FILE_FLAGS.get(unit, []), *flags, "-I", B.INCLUDE, "-c", os.path.join(B.CSRC, unit), "-o", obj]
    subprocess.run(cmd, check=True)
    B.check_device_asm(obj) if os.path.exists(obj[:-2] + f"-hip-amdgcn-amd-amdhsa-{B.ARCH}.s") else None
    objs = [os.path.join(B.OBJ, os.path.basename(s)[:-4] + ".o") for s in B.sources() if os.path.basename(s) != unit] + [obj]
    lib = os.path.join(out, f"libwfl_{name}.so")
    subprocess.run([hipcc, "-shared", "-fPIC", f"--offload-arch={B.ARCH}", *objs, "-o", lib], check=True)
    print(lib)


if __name__ == "__main__":
    main()

"""Compile the HIP kernels + C ABI into wfl-asr_amd/libwfl_asr_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(PKG, "libwfl_asr_hip.so")
INCLUDE = os.path.join(os.path.dirname(PKG), "include")
ARCH = "gfx950"
# -amdgpu-mfma-vgpr-form: keep MFMA accumulators in VGPRs (gfx950's register file is unified).  With the default AGPR
# form hipcc (ROCm 7.2) rotates the 64 accumulator registers of the software-pipelined GEMM loop through
# v_accvgpr_mov every iteration (~40 extra instructions per K tile).
FLAGS = ["-O3", "-fPIC", "-std=c++17", f"--offload-arch={ARCH}", "-Wno-unused-result", "-mllvm", "-amdgpu-mfma-vgpr-form=1"]
# -save-temps=obj leaves the device assembly next to each object; check_device_asm() reads it (below).
FLAGS += ["-save-temps=obj"]
# wavlm.hip: no SLP vectoriser, i.e. no compiler-made packed-f32 instructions in conv0 (the note at conv0_group_kernel)
PER_FILE_FLAGS = {"wavlm.hip": ["-fno-slp-vectorize"]}

# Packed-f32 VALU instructions whose LOW lane takes the HIGH half of src1 (op_sel[1] = 1) return a wrong low half in lanes 48-63, now and
# then, while another wave of the SIMD issues MFMAs (gfx950, found in round 3: tools/micro/conv0_probe.hip).  No
# object of this library may contain one: the build fails instead of shipping a kernel that is wrong only beside a neighbour.
import re
_BAD_PK = re.compile(r"^\s*v_pk_(fma|mul|add|max|min)_f32\b.*\bop_sel:\[[01],1")


def check_device_asm(obj: str) -> None:
    base = obj[:-2]
    asm = f"{base}-hip-amdgcn-amd-amdhsa-{ARCH}.s"
    if not os.path.exists(asm):
        raise RuntimeError(f"{asm} is missing: cannot check the device code of {obj} (was it built without -save-temps=obj?)")
    with open(asm, "r", errors="replace") as f:
        for ln, line in enumerate(f, 1):
            if _BAD_PK.match(line):
                raise RuntimeError(f"{asm}:{ln}: {line.strip()}\n  a packed-f32 instruction with op_sel[1] = 1 (src1's high half feeding the low "
                                   "lane) is not safe beside MFMA waves on gfx950; compile the file with -fno-slp-vectorize or restructure the code")


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps + [os.path.abspath(__file__)])


def build_library(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers += [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith(".h")]
    hipcc = _hipcc()
    jobs = []
    objs = []
    for src in sources():
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            jobs.append([hipcc, *FLAGS, *PER_FILE_FLAGS.get(os.path.basename(src), []), "-I", INCLUDE, "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed: {' '.join(cmd)}\n{r.stdout}\n{r.stderr}")

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(run, jobs))
    for obj in objs:
        check_device_asm(obj)
        for f in os.listdir(OBJ):        # keep the object and the device assembly; drop -save-temps' other by-products (they would travel)
            if f.startswith(os.path.basename(obj)[:-2] + "-") or f.startswith(os.path.basename(obj)[:-2] + ".hip-"):
                if not f.endswith(f"-{ARCH}.s"):
                    os.remove(os.path.join(OBJ, f))
    if force or jobs or _stale(LIB, objs):
        run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", LIB])
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))

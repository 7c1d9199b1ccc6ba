#!/usr/bin/env python3
"""Diagnostic (not product): gemm_mx.hip on the four encoder GEMM shapes of BASELINE configs[4] (Whisper-large-v3, 32 x 30 s per GPU), built
with -DWFL_GEMM_STAMPS (+ extra -D flags per variant), timed with HIP events, with the kernel's own cycle accounting per wave group:
prologue / epilogues / DMA waits / barriers / MFMA issue / L slot.  Usage: mx_lab.py name=-DFLAG,... [name2=...]; LAB_BUILD_ONLY=1 builds."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tools", "_diag")
SRC = os.path.join(ROOT, "wfl-asr_amd", "csrc")

DRV = '''
#include "common.h"
int g_wfl_gemm_kernel_id = 0;
extern "C" int diag_mx(const void* A, const void* Alo, long lda, const void* W, const float* wscale, const float* ascale, int M, int N, int K,
                       int P, int T, void* Cout, long ldc, long c_lead, const float* bias, const void* res, const void* res_lo, void* c_lo,
                       int act, void* c8, void* c8lo, long ldc8, unsigned long long* stamps, unsigned* err, void* stream) {
  GemmArgs g{};
  g.A = (const bf16_t*)A; g.a8_lo = (const unsigned char*)Alo; g.a8 = Alo ? 3 : 2; g.lda = lda; g.cin = K; g.W = (const bf16_t*)W;
  g.w8_scale = wscale; g.a8_scale = ascale; g.a8_static = 0.125f; g.a8_lead = 0;
  g.M = M; g.N = N; g.K = K; g.n_valid = N; g.P = P; g.T = T; g.C = Cout; g.ldc = ldc; g.c_lead = c_lead; g.c_pitch = P;
  g.bias = bias; g.res = (const bf16_t*)res; g.res_lo = (const bf16_t*)res_lo; g.c_lo = (bf16_t*)c_lo; g.ldres = ldc; g.alpha = 1.f; g.act = act;
  g.c8 = (unsigned char*)c8; g.c8_lo = (unsigned char*)c8lo; g.ldc8 = ldc8; g.c8_inv_scale = 8.f; g.err = err; g.stamps = stamps;
  return wfl_launch_gemm_mx(g, (hipStream_t)stream);
}
'''


def build(name, flags):
    os.makedirs(OUT, exist_ok=True)
    lib = os.path.join(OUT, f"libmx_lab_{name}.so")
    drv = os.path.join(OUT, "drv_mx.hip")
    open(drv, "w").write(DRV)
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "-std=c++17", "--offload-arch=gfx950", "-DWFL_GEMM_STAMPS", "-mllvm",
           "-amdgpu-mfma-vgpr-form=1", "-I", SRC, *flags, "-shared", os.path.join(SRC, "gemm_mx.hip"), drv, "-o", lib]
    subprocess.run(cmd, check=True)
    return lib


# name, K, N, act, residual, e4m3 output
SHAPES = [("qkv", 1280, 3840, 0, False, False), ("out+res", 1280, 1280, 0, True, False), ("fc1 gelu->e4m3", 1280, 5120, 1, False, True),
          ("fc2+res", 5120, 1280, 0, True, False)]


def main():
    variants = []
    for a in sys.argv[1:] or ["stock="]:
        name, _, fl = a.partition("=")
        variants.append((name, [f for f in fl.split(",") if f]))
    if os.environ.get("LAB_BUILD_ONLY"):
        for name, fl in variants:
            print(build(name, fl))
        return
    libs = {}
    for name, fl in variants:
        path = os.path.join(OUT, f"libmx_lab_{name}.so")
        if not os.path.exists(path):
            path = build(name, fl)
        lib = C.CDLL(path)
        lib.diag_mx.argtypes = [C.c_void_p] * 2 + [C.c_long] + [C.c_void_p] * 3 + [C.c_int] * 5 + [C.c_void_p, C.c_long, C.c_long] + \
                               [C.c_void_p] * 4 + [C.c_int, C.c_void_p, C.c_void_p, C.c_long, C.c_void_p, C.c_void_p, C.c_void_p]
        libs[name] = lib
    B, T, P, lead = int(os.environ.get("LAB_B", "32")), 1500, 1520, 16
    M = B * P
    R = lead + M + 256
    dev = "cuda"
    err = torch.zeros(64, dtype=torch.int32, device=dev)
    stamps = torch.zeros(256 * 2 * 8, dtype=torch.int64, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    for sname, K, N, act, res, out8 in SHAPES:
        A = torch.randint(0, 256, (R + 64, K), dtype=torch.uint8, device=dev)
        A[(A & 0x7f) == 0x7f] = 0x38                          # no NaN bytes
        Alo = torch.randint(0, 120, (R + 64, K), dtype=torch.uint8, device=dev)
        W = torch.randint(0, 120, (N, K), dtype=torch.uint8, device=dev)
        ws = torch.full((N,), 1e-3, device=dev)
        asc = torch.full((R + 64,), 1e-2, device=dev)
        Cb = torch.zeros(R, N, dtype=torch.bfloat16, device=dev)
        Clo = torch.zeros(R, N, dtype=torch.bfloat16, device=dev) if res else None
        Rs = torch.randn(R, N, device=dev).to(torch.bfloat16) if res else None
        Rlo = torch.zeros(R, N, dtype=torch.bfloat16, device=dev) if res else None
        C8 = torch.zeros(R, N, dtype=torch.uint8, device=dev) if out8 else None
        C8lo = torch.zeros(R, N, dtype=torch.uint8, device=dev) if out8 else None
        bias = torch.zeros(N, device=dev)
        flops = 2.0 * M * N * K
        for pair in (False, True):
            for vname, lib in libs.items():
                def call():
                    return lib.diag_mx(p(A[lead:]), p(Alo[lead:]) if pair else None, K, p(W), p(ws), p(asc), M, N, K, P, T, p(Cb), N, lead, p(bias),
                                       p(Rs), p(Rlo), p(Clo), act, p(C8), p(C8lo) if pair else None, N, p(stamps), p(err),
                                       C.c_void_p(torch.cuda.current_stream().cuda_stream))
                rc = call()
                if rc:
                    print(sname, vname, "launch failed", rc)
                    continue
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                n = 10
                e0.record()
                for _ in range(n):
                    call()
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 1e3 / n
                st = stamps.view(256, 2, 8).cpu().numpy().astype(np.float64)
                mt = 6 if pair else 5
                tiles = ((M + mt * 32 - 1) // (mt * 32)) * (N // 256)
                line = "%-16s %-6s %-8s %7.1f us %6.0f TF/s  tiles %5d (%.2f rounds)" % (sname, "pair" if pair else "single", vname, us, flops / us / 1e6, tiles, tiles / 256)
                for gq in (0, 1):
                    m = st[:, gq, :].mean(axis=0)
                    line += "\n      group %d: total %7.0f cyc | prologue %5.0f  epilogues %6.0f  L slot %6.0f  dma-wait %6.0f  barrier %6.0f  mfma %6.0f | steps %.0f" % (
                        gq, m[1], m[0], m[2], m[6], m[3], m[4], m[5], m[7])
                print(line, flush=True)


if __name__ == "__main__":
    main()

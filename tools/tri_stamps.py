#!/usr/bin/env python3
"""Diagnostic (not product): where a block of gemm256.hip's slice-by-slice walk (template TRI: the exact-label GEMM) spends its time.
Builds of gemm256.hip with -DWFL_GEMM_STAMPS and extra -D flags (name=-DFLAG,...; LAB_BUILD_ONLY=1 builds and stops), cfg2's shapes,
per block: setup / K loop / epilogue from the in-kernel stamps, next to the launch's wall time.
usage: tri_stamps.py stock= nostage=-DWFL_ABL_NOSTAGE nomma=-DWFL_ABL_NOMMA"""
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
int wfl_launch_gemm256(const GemmArgs& a, hipStream_t s);
int g_wfl_gemm_kernel_id = 0;
extern "C" int diag_split(const void* A, long seg_off, long lda, const void* W3, int M, int N, int K, int P, int T, void* Cout, void* Clo,
                          long ldc, long c_lead, const float* bias, const void* res, const void* res_lo, int act, int glu, int tri,
                          unsigned long long* stamps, void* stream) {
  GemmArgs g{};
  g.A = (const bf16_t*)A; g.lda = lda; g.cin = K; g.tap_stride = 0; g.tap_wrap = 1; g.seg_off = seg_off; g.W = (const bf16_t*)W3;
  g.M = M; g.N = N; g.K = 3 * K; g.n_valid = N; g.P = P; g.T = T; g.C = Cout; g.c_lo = (bf16_t*)Clo; g.ldc = ldc; g.c_lead = c_lead; g.c_pitch = P;
  g.bias = bias; g.res = (const bf16_t*)res; g.res_lo = (const bf16_t*)res_lo; g.ldres = ldc; g.alpha = 1.f; g.act = act; g.glu = glu;
  g.stamps = stamps;
  if (!tri) g.tap_wrap = 1;
  return wfl_launch_gemm256(g, (hipStream_t)stream);
}
'''


def build(name, flags):
    os.makedirs(OUT, exist_ok=True)
    lib = os.path.join(OUT, f"libtri_{name}.so")
    drv = os.path.join(OUT, "drv_tri.hip")
    open(drv, "w").write(DRV)
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "-std=c++17", "--offload-arch=gfx950", "-DWFL_GEMM_STAMPS", "-mllvm",
           "-amdgpu-mfma-vgpr-form=1", "-I", SRC, *flags, "-shared", os.path.join(SRC, "gemm256.hip"), drv, "-o", lib]
    subprocess.run(cmd, check=True)
    return lib


SHAPES = [("qkv", 512, 1536, 0, 0, 0), ("out_proj+res", 512, 512, 0, 0, 1), ("fc1+gelu", 512, 2048, 1, 0, 0), ("fc2+res", 2048, 512, 0, 0, 1)]


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
        path = os.path.join(OUT, f"libtri_{name}.so")
        if not os.path.exists(path):
            path = build(name, fl)
        lib = C.CDLL(path)
        lib.diag_split.argtypes = [C.c_void_p, C.c_long, C.c_long, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_long, C.c_long, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        libs[name] = lib
    B, T, P, lead = 16, 1500, 1520, 16
    M = B * P
    R = lead + M + 256
    for name, K, N, act, glu, has_res in SHAPES:
        A = (torch.randn(2 * R, K, device="cuda") * 0.5).to(torch.bfloat16)
        W3 = (torch.randn(N, 3 * K, device="cuda") * K ** -0.5).to(torch.bfloat16)
        Cb = torch.zeros(2 * R, N, dtype=torch.bfloat16, device="cuda")
        Rs = torch.randn(2 * R, N, device="cuda").to(torch.bfloat16)
        bias = torch.randn(N, device="cuda")
        stamps = torch.zeros(4096 * 8, dtype=torch.int64, device="cuda")
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

        def run(lib, with_stamps):
            return lib.diag_split(C.c_void_p(A.data_ptr() + lead * K * 2), R * K, K, C.c_void_p(W3.data_ptr()), M, N, K, P, T,
                                  C.c_void_p(Cb.data_ptr()), C.c_void_p(Cb.data_ptr() + R * N * 2), N, lead, C.c_void_p(bias.data_ptr()),
                                  C.c_void_p(Rs.data_ptr()) if has_res else None, C.c_void_p(Rs.data_ptr() + R * N * 2) if has_res else None,
                                  act, glu, 1, C.c_void_p(stamps.data_ptr()) if with_stamps else None, st)
        print(f"--- {name}  K={K} N={N}")
        for n, lib in libs.items():
            for _ in range(300):                        # (the clock settles under load: warm launches first)
                assert run(lib, False) == 0
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    run(lib, False)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 100)
            stamps.zero_()
            assert run(lib, True) == 0
            torch.cuda.synchronize()
            s = stamps.view(-1, 8).cpu().numpy().astype(np.int64)
            s = s[s[:, 4] > 0]
            ph = np.diff(s[:, :5], axis=1) / 100.0
            us = np.median(ts)
            print(f"  {n:10s} {us:7.1f} us | blocks {len(s):4d} | setup {np.median(ph[:, 0]):5.2f} K loop {np.median(ph[:, 1]):6.2f} "
                  f"epi stage {np.median(ph[:, 2]):5.2f} store {np.median(ph[:, 3]):5.2f} | us per K slice {np.median(ph[:, 1]) / (K / 32):.3f} "
                  f"| K-loop clock {np.median((s[:, 6] - s[:, 5]) / np.maximum(s[:, 2] - s[:, 1], 1) / 10.0):.2f} GHz "
                  f"| block start spread {(s[:, 0].max() - s[:, 0].min()) / 100.0:6.1f} us, last end {(s[:, 4].max() - s[:, 0].min()) / 100.0:6.1f} us")


if __name__ == "__main__":
    main()

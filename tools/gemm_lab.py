#!/usr/bin/env python3
"""Diagnostic (not product): A/B builds of the GEMM translation units with extra -D flags, timed on the config-2 shapes in
ONE process (interleaved rounds).  Usage: gemm_lab.py name=-DFLAG1,-DFLAG2 name2= ...   (empty flag list = stock build)
Every build carries -DWFL_GEMM_STAMPS so the K-loop time per block (stamps 1 -> 2) is reported next to the wall time."""
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
extern "C" int diag_gemm(const void* A, long lda, int cin, long tap_stride, const void* W, int M, int N, int K, int P, int T,
                         void* Cout, long ldc, long c_lead, const float* bias, const void* res, int act, unsigned long long* stamps,
                         const float* ln_s, void* stream, const void* res_lo, void* c_lo, float* stats_out, const float* w8_scale) {
  GemmArgs g{};
  g.A = (const bf16_t*)A; g.lda = lda; g.cin = cin > 0 ? cin : K; g.tap_stride = tap_stride; g.W = (const bf16_t*)W;
  g.M = M; g.N = N; g.K = K; g.n_valid = N; g.P = P; g.T = T; g.C = Cout; g.ldc = ldc; g.c_lead = c_lead; g.c_pitch = P;
  g.bias = bias; g.res = (const bf16_t*)res; g.ldres = ldc; g.alpha = 1.f; g.act = act; g.stamps = stamps; g.ln_s = ln_s; g.ln_eps = 1e-5f;
  g.res_lo = (const bf16_t*)res_lo; g.c_lo = (bf16_t*)c_lo; g.stats_out = stats_out; g.w8_scale = w8_scale;
  return wfl_launch_gemm(g, (hipStream_t)stream);
}
'''


def build(name, flags):
    os.makedirs(OUT, exist_ok=True)
    lib = os.path.join(OUT, f"libgemm_lab_{name}.so")
    drv = os.path.join(OUT, "drv.hip")
    open(drv, "w").write(DRV)
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "-std=c++17", "--offload-arch=gfx950", "-DWFL_GEMM_STAMPS", "-mllvm",
           "-amdgpu-mfma-vgpr-form=1", "-I", SRC, *flags, "-shared", os.path.join(SRC, "gemm.hip"), os.path.join(SRC, "gemm256.hip"), os.path.join(SRC, "gemm_stream.hip"), os.path.join(SRC, "gemm_mx.hip"),
           drv, "-o", lib]
    subprocess.run(cmd, check=True)
    return lib


SHAPES = [("out_proj+res", 512, 512, 0, True), ("qkv-like", 512, 1536, 0, False), ("fc1 gelu", 512, 2048, 1, False),
          ("qkvLN", 512, 1536, 0, False), ("fc1LN gelu", 512, 2048, 1, False),
          ("fc2+res", 2048, 512, 0, True), ("k31 conv gelu", 15872, 512, 1, False), ("ff1_b", 1024, 512, 0, True)]


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
        path = os.path.join(OUT, f"libgemm_lab_{name}.so")
        if not os.path.exists(path):
            path = build(name, fl)
        lib = C.CDLL(path)
        lib.diag_gemm.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_long, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_void_p, C.c_long, C.c_long, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        libs[name] = lib
    B, T, P, lead = (32 if os.environ.get("LAB_CFG5") else 16), 1500, 1520, 16
    w8 = bool(os.environ.get("LAB_W8"))                # e4m3 weights + per-channel scales (gemm_stream's W8 instantiations)
    M = B * P
    R = lead + M + 256
    only = os.environ.get("LAB_SHAPES")
    shapes = SHAPES if not os.environ.get("LAB_CFG5") else [("qkv", 1280, 3840, 0, False), ("out+res", 1280, 1280, 0, True),
                                                            ("fc1 gelu", 1280, 5120, 1, False), ("fc2+res", 5120, 1280, 0, True)]
    for name, K, N, act, res in shapes:
        if only and name.split()[0] not in only.split(","):
            continue
        kin = K if os.environ.get("LAB_CFG5") else min(K, 2048)
        A = (torch.randn(R + 64, kin, device="cuda") * 0.5).to(torch.bfloat16)
        W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
        Cb = torch.zeros(R, N, dtype=torch.bfloat16, device="cuda")
        Rs = (torch.randn(R, N, device="cuda")).to(torch.bfloat16)
        lo_mode = bool(os.environ.get("LAB_LO")) and res            # residual stream as hi + lo (+ statistics with LAB_STATS)
        Rlo = (torch.randn(R, N, device="cuda") * 0.003).to(torch.bfloat16) if lo_mode else None
        Clo = torch.zeros(R, N, dtype=torch.bfloat16, device="cuda") if lo_mode else None
        Sts = torch.zeros(R * 8, dtype=torch.float32, device="cuda") if (res and os.environ.get("LAB_STATS") and N <= 1024) else None
        bias = torch.randn(N, device="cuda")
        stamps = torch.zeros(4096 * 8, dtype=torch.int64, device="cuda")
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        lda = kin if (K <= 2048 or os.environ.get("LAB_CFG5")) else 512
        tapped = K > 2048 and not os.environ.get("LAB_CFG5")
        W8b = W.to(torch.float8_e4m3fn) if w8 else None
        w8s = torch.ones(N, device="cuda") if w8 else None
        Wref = W8b.float() if w8 else W.float()
        aoff = lead * lda * 2
        fl = 2.0 * B * T * N * K
        ln = "LN" in name
        ln_s = W.float().sum(1).contiguous()

        def run(lib, with_stamps):
            return lib.diag_gemm(C.c_void_p(A.data_ptr() + aoff), lda, 512 if tapped else 0, 512 if tapped else 0, C.c_void_p((W8b if w8 else W).data_ptr()), M, N, K, P, T,
                                 C.c_void_p(Cb.data_ptr()), N, lead, C.c_void_p(bias.data_ptr()),
                                 C.c_void_p(Rs.data_ptr()) if res else None, act,
                                 C.c_void_p(stamps.data_ptr()) if with_stamps else None,
                                 C.c_void_p(ln_s.data_ptr()) if ln else None, st,
                                 C.c_void_p(Rlo.data_ptr()) if lo_mode else None, C.c_void_p(Clo.data_ptr()) if lo_mode else None,
                                 C.c_void_p(Sts.data_ptr()) if Sts is not None else None,
                                 C.c_void_p(w8s.data_ptr()) if w8 else None)

        times = {n: [] for n in libs}
        # independent reference for 64 rows spread over the run (fp32 matmul of the bf16 operands)
        ridx = torch.arange(0, 64, device="cuda") * 379 % (B * P)
        ridx = ridx[(ridx % P) < T]
        Arows = torch.as_strided(A.view(-1), (B * P, K), (lda, 1), lead * lda)[ridx].float()
        if ln:
            Arows = torch.nn.functional.layer_norm(Arows, (K,))
        ref = Arows @ Wref.T + bias
        if act == 1:
            ref = torch.nn.functional.gelu(ref)
        if res:
            ref = ref + Rs[lead + ridx].float()
            if lo_mode:
                ref = ref + Rlo[lead + ridx].float()
        for rnd in range(5):
            for n, lib in libs.items():
                for _ in range(2):
                    assert run(lib, False) == 0
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    run(lib, False)
                e1.record()
                torch.cuda.synchronize()
                times[n].append(e0.elapsed_time(e1) * 100)
        print(f"--- {name}  K={K} N={N}")
        first_out = None
        for n, lib in libs.items():
            stamps.zero_()
            Cb.zero_()
            assert run(lib, True) == 0
            torch.cuda.synchronize()
            # every build against the first one, whole output, over several launches (a race in a build shows as a difference)
            if first_out is None:
                first_out = Cb.clone()
            else:
                nd = 0
                for _ in range(8):
                    Cb.zero_()
                    run(lib, False)
                    torch.cuda.synchronize()
                    nd += int((Cb.view(torch.int16) != first_out.view(torch.int16)).sum().item())
                print(f"  {n:14s} elements differing from the first build's output over 8 launches: {nd}")
            out = Cb[lead + ridx].float()
            diff = ((out - ref).abs() / (ref.abs() + 1.0)).max().item()
            s = stamps.view(-1, 8).cpu().numpy().astype(np.int64)
            s = s[s[:, 4] > 0]
            ph = np.diff(s[:, :5], axis=1) / 100.0
            us = np.median(times[n])
            print(f"  {n:14s} {us:7.1f} us (min {min(times[n]):7.1f}) {fl / us / 1e6:6.0f} TF | blocks {len(s):4d} | setup {np.median(ph[:, 0]):5.2f} "
                  f"K loop {np.median(ph[:, 1]):6.2f} epi stage {np.median(ph[:, 2]):5.2f} store {np.median(ph[:, 3]):5.2f} | "
                  f"us/kstep32 {np.median(ph[:, 1]) / (K / 32):.3f} | K-loop clock {np.median((s[:, 6] - s[:, 5]) / np.maximum(s[:, 2] - s[:, 1], 1) / 10.0):.2f} GHz "
                  f"| max rel-ish err vs torch {diff:.3g}")


if __name__ == "__main__":
    main()

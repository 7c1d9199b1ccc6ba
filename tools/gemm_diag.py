#!/usr/bin/env python3
"""Diagnostic (not product): build the GEMM with -DWFL_GEMM_STAMPS into tools/_diag/libgemm_diag.so, run the shapes of
the config-2 forward and print, per shape, wall time plus the per-block phase breakdown from 100 MHz in-kernel stamps
(0 entry, 1 first tile landed, 2 K loop done, 3 epilogue staged, 4 stored) and the blocks-per-CU residency."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tools", "_diag")
SRC = os.path.join(ROOT, "wfl-asr_amd", "csrc")


def build(nstage=4):
    os.makedirs(OUT, exist_ok=True)
    lib = os.path.join(OUT, f"libgemm_diag_s{nstage}.so")
    drv = os.path.join(OUT, "drv.hip")
    open(drv, "w").write('''
#include "common.h"
extern "C" int diag_gemm(const void* A, long lda, int cin, long tap_stride, const void* W, int M, int N, int K, int P, int T,
                         void* Cout, long ldc, long c_lead, const float* bias, const void* res, int act, unsigned long long* stamps,
                         void* stream) {
  GemmArgs g{};
  g.A = (const bf16_t*)A; g.lda = lda; g.cin = cin > 0 ? cin : K; g.tap_stride = tap_stride; g.W = (const bf16_t*)W;
  g.M = M; g.N = N; g.K = K; g.n_valid = N; g.P = P; g.T = T; g.C = Cout; g.ldc = ldc; g.c_lead = c_lead; g.c_pitch = P;
  g.bias = bias; g.res = (const bf16_t*)res; g.ldres = ldc; g.alpha = 1.f; g.act = act; g.stamps = stamps;
  return wfl_launch_gemm(g, (hipStream_t)stream);
}
''')
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-fPIC", "-std=c++17", "--offload-arch=gfx950", "-DWFL_GEMM_STAMPS", f"-DNSTAGE={nstage}", "-mllvm", "-amdgpu-mfma-vgpr-form=1", "-I", SRC,
           "-shared", os.path.join(SRC, "gemm.hip"), os.path.join(SRC, "gemm256.hip"), drv, "-o", lib]
    subprocess.run(cmd, check=True)
    return lib


def main(nstage):
    print(f"==== NSTAGE={nstage}")
    lib = C.CDLL(build(nstage))
    lib.diag_gemm.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_long, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                              C.c_void_p, C.c_long, C.c_long, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    B, T, P, lead = 16, 1500, 1520, 16
    M = B * P
    R = lead + M + 256
    shapes = [("out_proj+res", 512, 512, 0, True), ("qkv-like", 512, 1536, 0, False), ("fc1 gelu", 512, 2048, 1, False),
              ("fc2+res", 2048, 512, 0, True), ("k31 conv gelu", 15872, 512, 1, False), ("ff2", 1024, 512, 0, True)]
    for name, K, N, act, res in shapes:
        kin = min(K, 2048)
        A = (torch.randn(R + 64, kin, device="cuda") * 0.5).to(torch.bfloat16)
        W = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
        Cb = torch.zeros(R, N, dtype=torch.bfloat16, device="cuda")
        bias = torch.randn(N, device="cuda")
        big = os.environ.get("WFL_GEMM_TILE", "256") != "128" and N % 256 == 0
        tiles = ((M + 255) // 256) * (N // 256) if big else ((M + 127) // 128) * (N // 128)
        stamps = torch.zeros(tiles * 8, dtype=torch.int64, device="cuda")
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        # conv-like addressing for K > kin: lda = 512 with contiguous taps (k31 conv)
        lda = kin if K <= 2048 else 512
        aoff = lead * lda * 2

        def run(with_stamps):
            return lib.diag_gemm(C.c_void_p(A.data_ptr() + aoff), lda, 0, 0, C.c_void_p(W.data_ptr()), M, N, K, P, T,
                                 C.c_void_p(Cb.data_ptr()), N, lead, C.c_void_p(bias.data_ptr()),
                                 C.c_void_p(Cb.data_ptr()) if res else None, act,
                                 C.c_void_p(stamps.data_ptr()) if with_stamps else None, st)
        for _ in range(3):
            assert run(False) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run(False)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        assert run(True) == 0
        torch.cuda.synchronize()
        s = stamps.view(tiles, 8).cpu().numpy().astype(np.int64)
        t0 = s[:, 0].min()
        rel = (s[:, :5] - t0) / 100.0                     # us
        dur = rel[:, 4] - rel[:, 0]
        ph = np.diff(rel[:, :5], axis=1)
        hw = s[:, 7]
        cu = (s[:, 6] << 16) | (hw & 0xFFFF00)            # xcc | se/sh/cu bits (wave slot bits dropped)
        # residency: max number of blocks alive at once on one (xcc, cu id)
        keys = {}
        for i in range(tiles):
            keys.setdefault(int(cu[i]), []).append((rel[i, 0], rel[i, 4]))
        maxres = 0
        for k, iv in keys.items():
            ev = sorted([(a, 1) for a, b in iv] + [(b, -1) for a, b in iv])
            c = m = 0
            for _, d in ev:
                c += d
                m = max(m, c)
            maxres = max(maxres, m)
        fl = 2.0 * B * T * N * K
        print(f"{name:16s} K={K:5d} N={N:4d} tiles={tiles:5d}  {us:7.1f} us  {fl / us / 1e6:6.0f} TF | block total med {np.median(dur):6.2f} us "
              f"| phases med us: setup+1st tile {np.median(ph[:, 0]):5.2f}, K loop {np.median(ph[:, 1]):6.2f}, epi stage {np.median(ph[:, 2]):5.2f}, "
              f"store {np.median(ph[:, 3]):5.2f} | last block ends {rel[:, 4].max():7.1f} us | distinct CU keys {len(keys)} max resident/CU {maxres}")
        starts = np.sort(rel[:, 0])
        print(f"      block start times us: p50 {np.percentile(starts, 50):.1f} p75 {np.percentile(starts, 75):.1f} p90 {np.percentile(starts, 90):.1f} max {starts.max():.1f}")


if __name__ == "__main__":
    for ns in (int(a) for a in (sys.argv[1:] or ["3", "4"])):
        main(ns)

#!/usr/bin/env python3
"""Diagnostic (not product): per-launch time of the exact-label GEMM (wfl_op_gemm_split) at cfg2's shapes, in the walk the environment
selects -- WFL_TRI=0 segment-major everywhere, WFL_TRI_RES=0 slice-by-slice except the residual launches, default slice-by-slice.
usage: tri_lab.py            (run it once per environment; same box)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import wfl_asr_amd  # noqa: E402,F401
import gpu_util as G  # noqa: E402
from wfl_asr_amd import _lib  # noqa: E402

SHAPES = [("qkv", 512, 1536, 0, 0, 0), ("out_proj+res", 512, 512, 0, 0, 1), ("fc1+gelu", 512, 2048, 1, 0, 0), ("fc2+res", 2048, 512, 0, 0, 1),
          ("glu", 512, 1024, 0, 1, 0)]


def main():
    B, T = 16, 1500
    print("WFL_TRI=%s WFL_TRI_RES=%s" % (os.environ.get("WFL_TRI", "1"), os.environ.get("WFL_TRI_RES", "1")))
    for name, K, N, act, glu, has_res in SHAPES:
        rows = G.Rows(B, T, K)
        both = (torch.randn(2 * rows.R, K, device="cuda") * 0.5).to(torch.bfloat16)
        w3 = (torch.randn(N, 3 * K, device="cuda") / K ** 0.5).to(torch.bfloat16)
        n_out = N // 2 if glu else N
        o = G.Rows(B, T, n_out)
        oo = torch.zeros(2 * o.R, n_out, dtype=torch.bfloat16, device="cuda")
        rr = torch.randn(2 * o.R, n_out, device="cuda").to(torch.bfloat16) if has_res else None
        bias = torch.zeros(N, device="cuda")

        def call():
            rc = G.lib().wfl_op_gemm_split(G.ptr(both, rows.lead * K), G.ptr(both, (rows.R + rows.lead) * K), K, 0, 0, G.ptr(w3), B * rows.P, N, K, N,
                                           rows.P, T, G.ptr(oo), G.ptr(oo, o.R * n_out), n_out, o.lead, o.P, G.ptr(bias),
                                           G.ptr(rr) if has_res else None, G.ptr(rr, o.R * n_out) if has_res else None, n_out, 1.0, act, glu,
                                           G.stream())
            _lib.check(rc, "wfl_op_gemm_split")
        for _ in range(5):
            call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 50
        e0.record()
        for _ in range(reps):
            call()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / reps
        fl = 3 * 2.0 * B * T * N * K
        print(f"  {name:14s} K={K:5d} N={N:5d}  {us:8.1f} us  {fl / us * 1e-6:7.1f} TFLOP/s of bf16 MFMA work")


if __name__ == "__main__":
    main()

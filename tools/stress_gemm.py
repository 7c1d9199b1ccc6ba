#!/usr/bin/env python3
"""Diagnostic (not product): race / indexing screen for the streaming GEMM (gemm_stream.hip).  Random shapes, epilogues and
batch geometries, every output row checked against a torch fp32 reference of the bf16 operands, halos checked for zero; the
same launch is repeated and must be bit-identical.  A sync bug in the counted-vmcnt / ping-pong / ticket logic shows up as a
rare wrong tile, so this runs a few hundred launches."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch
import torch.nn.functional as F

import gpu_util as G


def bf(x):
    return x.to(torch.bfloat16).float()


def main(iters=240, seed=0):
    rng = np.random.default_rng(seed)
    worst = 0.0
    t0 = time.time()
    # a fixed pool of configurations (torch picks a matmul kernel per new shape, which takes seconds), each visited many times
    # with fresh data: a race needs repetition, not variety
    pool = []
    for i in range(16):
        mode = ["plain", "res", "ln", "conv"][i % 4]
        N = int(rng.choice([256, 512, 768, 1024, 1536, 2048]))
        T = int(rng.choice([49, 200, 499, 1500]))
        B = int(rng.integers(1, 17)) if T >= 499 else int(rng.integers(8, 65))
        act = int(rng.choice([0, 1, 2])) if mode in ("plain", "conv") else (int(rng.choice([0, 1])) if mode == "ln" else 0)
        if mode == "conv":
            C = int(rng.choice([64, 128, 256, 512]))
            k = int(rng.choice([3, 5, 31]))
            K = C * k
        else:
            C, k = 0, 0
            K = int(rng.choice([256, 512, 768, 1024, 2048]))
        pool.append((mode, N, T, B, act, C, k, K))
    for it in range(iters):
        mode, N, T, B, act, C, k, K = pool[it % len(pool)]
        g = torch.Generator(device="cuda").manual_seed(1000 + it)
        x0 = torch.randn(B, T, K if mode != "conv" else C, device="cuda", generator=g) * 0.7
        if mode == "ln":
            x0 = x0 * 2 + 0.5
        a = G.Rows(B, T, x0.shape[-1], halo=32, lead=32).set(x0)
        w = torch.randn(N, K, device="cuda", generator=g) * K ** -0.5
        bias = torch.randn(N, device="cuda", generator=g) * 0.1
        out = G.Rows(B, T, N, halo=32, lead=32)
        r0 = torch.randn(B, T, N, device="cuda", generator=g)
        if mode == "res":
            out.set(r0)
        M = B * a.P
        if mode == "ln":
            gamma = 1 + 0.2 * torch.randn(K, device="cuda", generator=g)
            beta = 0.1 * torch.randn(K, device="cuda", generator=g)
            wf = (w * gamma).to(torch.bfloat16)
            wp, bp = G.pad_weight(wf.float(), bias + w @ beta)
            call = lambda: G.gemm_ln(a.buf, a.lead * K, K, wp, M, N, a.P, T, out.buf, N, out.lead, out.P, bp, wf.float().sum(1).contiguous(), 1e-5, act)
            ref = F.layer_norm(a.get(), (K,), gamma, beta, 1e-5) @ w.T + bias
            tol = 3e-2
        elif mode == "conv":
            wc = w.view(N, k, C)                       # tap-major rows
            wp, bp = G.pad_weight(w, bias)
            left = (k - 1) // 2
            call = lambda: G.gemm(a.buf, (a.lead - left) * C, C, wp, M, N, a.P, T, out.buf, N, out.lead, out.P, bias=bp, act=act, cin=C, tap_stride=C)
            ref = F.conv1d(F.pad(bf(x0).transpose(1, 2), (left, k - 1 - left)), bf(wc).permute(0, 2, 1), bias).transpose(1, 2)
            tol = 2e-2
        else:
            wp, bp = G.pad_weight(w, bias)
            res = out.buf if mode == "res" else None
            call = lambda: G.gemm(a.buf, a.lead * K, K, wp, M, N, a.P, T, out.buf, N, out.lead, out.P, bias=bp, res=res, ldres=N, alpha=1.0, act=act)
            ref = a.get() @ bf(w).T + bias
            tol = 2e-2
        print(it, mode, 'B', B, 'T', T, 'K', K, 'N', N, 'act', act, flush=True)
        ref = [ref, F.gelu(ref), F.relu(ref)][act]
        if mode == "res":
            ref = bf(r0) + ref
        call()
        torch.cuda.synchronize()
        got = out.get()
        err = ((got - ref).abs() / (ref.abs() + 1.0)).max().item()
        assert out.halo_is_zero(), (it, mode, "halo written")
        assert err < tol, (it, mode, B, T, K, N, act, err)
        if mode != "res":                               # in-place residual is not idempotent
            first = out.buf.clone()
            for _ in range(3):
                call()
            torch.cuda.synchronize()
            assert torch.equal(first, out.buf), (it, mode, "non-deterministic")
        worst = max(worst, err)
    print(f"{iters} launches x (1 + repeats) ok, worst rel-ish err {worst:.4f}, {time.time() - t0:.1f} s")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 240)

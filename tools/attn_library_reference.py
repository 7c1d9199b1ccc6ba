#!/usr/bin/env python3
"""Diagnostic (not product, not used by it): torch's scaled_dot_product_attention (the library flash kernels of this image) on the
attention shapes of the cfg2 / cfg5 forward -- a yardstick for attention.hip."""
import torch
import torch.nn.functional as F

for name, B, H, T, D in [("whisper-base (cfg2)", 16, 8, 1500, 64), ("conformer heads (cfg2)", 16, 2, 1500, 256), ("whisper-large-v3 (cfg5)", 32, 20, 1500, 64)]:
    q, k, v = [(torch.randn(B, H, T, D, device="cuda") * 0.5).to(torch.bfloat16) for _ in range(3)]
    for backend in ("default",):
        try:
            for _ in range(3):
                o = F.scaled_dot_product_attention(q, k, v)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            n = 10
            for _ in range(n):
                o = F.scaled_dot_product_attention(q, k, v)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / n * 1e3
            print(f"{name:26s} B={B} H={H} T={T} D={D}: {us:8.1f} us  {4.0*B*H*T*T*D/us/1e6:6.0f} TFLOP/s")
        except Exception as e:
            print(name, "failed:", type(e).__name__, str(e)[:200])

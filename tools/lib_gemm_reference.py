#!/usr/bin/env python3
"""Diagnostic (not product, not used by it): what the vendor library (torch.nn.functional.linear -> hipBLASLt / rocBLAS, bf16, NO epilogue)
reaches on the GEMM shapes of the cfg2 / cfg5 forward on this GPU -- a yardstick for the hand-written kernels, which also fuse the
bias / LayerNorm fold / GELU / residual hi + lo / statistics epilogues and run the k = 31 conv without an im2col buffer."""
import torch, time
torch.backends.cuda.matmul.allow_tf32 = False
shapes = [("out_proj", 24000, 512, 512), ("qkv", 24000, 1536, 512), ("fc1", 24000, 2048, 512), ("fc2", 24000, 512, 2048), ("k31conv-as-gemm", 24000, 512, 15872),
          ("large-v3 fc1", 48000, 5120, 1280), ("large-v3 fc2", 48000, 1280, 5120)]
for name, M, N, K in shapes:
    a = (torch.randn(M, K, device="cuda") * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    for _ in range(3):
        c = torch.nn.functional.linear(a, w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 20
    for _ in range(n):
        c = torch.nn.functional.linear(a, w)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    print(f"{name:18s} M={M} N={N} K={K}: {us:8.1f} us  {2.0*M*N*K/us/1e6:7.0f} TFLOP/s")

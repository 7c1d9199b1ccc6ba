#!/usr/bin/env python3
"""Diagnostic (not product): time wfl_op_attention at the Conformer head sizes.  usage: attn_bench.py d heads B [T]
(run once per WFL_ATTN_VARIANT value to A/B the head_dim 256 / 384 kernels)"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import wfl_asr_amd  # noqa: F401,E402
from wfl_asr_amd import _lib  # noqa: E402

d, heads, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
T = int(sys.argv[4]) if len(sys.argv) > 4 else 1500
P, lead = T + 20, 16
R = lead + B * P + 256
lib = _lib.load()
qkv = (torch.randn(R, 3 * d, device="cuda") * 0.5).to(torch.bfloat16)
o = torch.zeros(R, d, dtype=torch.bfloat16, device="cuda")
p = lambda t, off=0: C.c_void_p(t.data_ptr() + off)
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def run():
    _lib.check(lib.wfl_op_attention(p(qkv), 3 * d, lead, p(qkv, 2 * d * 2), 3 * d, p(o), d, B, T, P, heads, d, s), "attn")


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
n = 10
for _ in range(n):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
fl = 4.0 * B * T * T * d
print(f"d {d} heads {heads} (head_dim {d // heads}) B {B} T {T} variant {os.environ.get('WFL_ATTN_VARIANT', '0')}: {ms * 1e3:.1f} us, {fl / ms / 1e9:.0f} TFLOP/s")

#!/usr/bin/env python3
"""Diagnostic (not product): Whisper forwards (cfg2 with 1 clip; the default head with 2 clips) on one HIP stream beside the attention
launches that disturbed conv0's group-norm kernel on another, compared bit for bit with their single-stream
results.  Round 2: 0 of 720 differ.  usage: soak_two_streams.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from wfl_asr_amd import _lib
import synthetic as synth
from wfl_asr_amd.tagger import BIOPhonemeTagger
lib = _lib.load()
def build(cfg, seed=1):
    labels = synth.make_labels(70)
    sd = synth.make_state_dict(cfg, len(labels), seed=seed)
    m = BIOPhonemeTagger(cfg, labels); m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); m.to("cuda")
    return m
p = lambda t, off=0: C.c_void_p(t.data_ptr() + off * t.element_size())
def sp(): return C.c_void_p(torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device="cuda").manual_seed(1)
def mk(*s, sc=0.5): return (torch.randn(*s, device="cuda", generator=g) * sc).to(torch.bfloat16)
lead = 72
def mk_attn(dd, hh, TT, n):
    PP = (TT + 72 + 7) // 8 * 8; RR = lead + PP + 256
    q = mk(RR, 3 * dd); oo = torch.zeros(RR, dd, dtype=torch.bfloat16, device="cuda")
    def f():
        for _ in range(n): _lib.check(lib.wfl_op_attention(p(q), 3 * dd, lead, p(q, 2 * dd), 3 * dd, p(oo), dd, 1, TT, PP, hh, dd, sp()), "a")
    return f
dists = {"attention hd 64 T 64 x12": mk_attn(768, 12, 64, 12), "attention hd 64 T 1500 x2": mk_attn(512, 8, 1500, 2)}
base = synth.make_clip(7000, 480000, seed=1) * 0.8
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
for name, cfg, B in (("Whisper-base cfg2, 1 clip", synth.baseline_config(1), 1), ("Whisper-base + default head, 2 clips", synth.base_config("whisper"), 2)):
    m = build(cfg)
    items = [torch.from_numpy(np.ascontiguousarray(np.stack([np.roll(base, 997 * i + 13 * j) for j in range(B)])).astype(np.float32)).cuda() for i in range(12)]
    lang = torch.zeros(B, dtype=torch.int32, device="cuda")
    ref = [m.label(x, lang, threshold=0.5, want_logits=True).logits.clone() for x in items]
    torch.cuda.synchronize()
    for dn, d in dists.items():
        bad = 0
        for rep in range(15):
            outs = []
            for k, x in enumerate(items):
                with torch.cuda.stream(s1):
                    d()
                with torch.cuda.stream(s0):
                    outs.append(m.label(x, lang, threshold=0.5, want_logits=True))
            torch.cuda.synchronize()
            bad += sum(not torch.equal(o.logits, ref[k]) for k, o in enumerate(outs))
        print("soak:", name, "beside", dn, "->", bad, "of 180 differ")

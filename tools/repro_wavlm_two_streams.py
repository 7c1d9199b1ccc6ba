#!/usr/bin/env python3
"""Diagnostic (not product), OPEN ISSUE: two WavLM forwards running concurrently on two HIP streams (different model instances,
different workspaces) now and then change each other's result -- the encoder output of a WavLM-base forward differs from its
single-stream value in 15-40 % of the runs while another WavLM-base forward runs beside it; the same forwards alternating on ONE
stream are bit-exact.  Established so far (round 2): the first differences already show in the feature encoder
(a small chance per kernel that grows along the forward, not one faulty kernel); the attention and GEMM kernels are bit-exact under
the same concurrency when driven directly through wfl_op_* (also with the producer -> consumer chain and large scores); no forward
writes outside its workspace (guard bands) or reads stale workspace contents (poisoned workspaces); Whisper forwards (B = 1 and
B = 16, eager and graph replay) do not show it; a neighbour that stops before its own attention does not disturb.  Mitigation in the product: WavLM / mel models keep ONE forward on the GPU at a
time (infer.py:_forward_items_by_length, bench.py --inflight default).  usage: repro_wavlm_two_streams.py"""
import os, sys, dataclasses
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from wfl_asr_amd import synth
from wfl_asr_amd.archs import WAVLM
from wfl_asr_amd.tagger import BIOPhonemeTagger
def build(cfg, seed=1, nph=70):
    labels = synth.make_labels(nph)
    sd = synth.make_state_dict(cfg, len(labels), seed=seed)
    m = BIOPhonemeTagger(cfg, labels); m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}); m.to("cuda")
    return m
def wl(idx, layers, **over):
    cfg = synth.baseline_config(idx)
    a = dataclasses.asdict(WAVLM["base" if idx == 0 else "large"]); a["layers"] = layers; a.update(over)
    cfg["model"]["wavlm_model"] = "local/wavlm-x"; cfg["model"]["encoder_arch"] = a
    return cfg
m = build(wl(0, 1)); nb = build(wl(0, 1), seed=2, nph=50)
base = synth.make_clip(7000, 480000, seed=1) * 0.8
items = [torch.from_numpy(np.ascontiguousarray(np.roll(base, 997 * i)[:100000]).astype(np.float32)[None]).cuda() for i in range(16)]
ref = [m.encode(x).clone() for x in items]
torch.cuda.synchronize()
s0 = torch.cuda.Stream()
bad = 0
for rep in range(6):
    outs = []
    with torch.cuda.stream(s0):
        for k, x in enumerate(items):
            nb.encode(items[(k + 5) % 16])
            outs.append(m.encode(x))
            nb.encode(items[(k + 7) % 16])
    torch.cuda.synchronize()
    bad += sum(not torch.equal(o, ref[k]) for k, o in enumerate(outs))
print("victim and neighbour ALTERNATING ON ONE STREAM (no concurrency): victim differs in", bad, "of 96")
s1 = torch.cuda.Stream()
bad = 0
for rep in range(6):
    outs = []
    for k, x in enumerate(items):
        with torch.cuda.stream(s1):
            for _ in range(3): nb.encode(items[(k + 5) % 16])
        with torch.cuda.stream(s0):
            outs.append(m.encode(x))
    torch.cuda.synchronize()
    bad += sum(not torch.equal(o, ref[k]) for k, o in enumerate(outs))
print("two streams: victim differs in", bad, "of 96")

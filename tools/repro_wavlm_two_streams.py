#!/usr/bin/env python3
"""Diagnostic (not product): a WavLM-base encoder forward on one HIP stream beside other WavLM-base forwards on a second stream
(another model instance, separate workspaces), compared bit for bit with its single-stream result; and the same two forwards
alternating on ONE stream.  Round 2 found 15-40 % of the two-stream runs different: conv0's group-norm kernel produced wrong rows
when its workgroups shared a CU with another forward's attention workgroups.  Round 3 found the cause with tools/micro/conv0_probe.hip
(a packed-f32 instruction form that is unsafe beside another wave's MFMAs: DESIGN.md section 7); with conv0 compiled without it both
lines read 0 of 96.
usage: repro_wavlm_two_streams.py"""
import os, sys, dataclasses
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import synthetic as synth
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

#!/usr/bin/env python3
"""Diagnostic (not product): what the waveforms' PCIe crossing costs the cfg2 step, and which part of it -- the copy itself beside the
forwards, the events that order it, or the stream it is issued from.  Same model / batch / loop as bench.py's with-H2D leg.
modes: resident | copy_unused (copies run, forwards read the resident batch) | events_only | copy_stream (bench.py's leg) |
       same_stream (copy on the forward's own stream, as Labeler._run_batches does) | copy_stream_ahead2 (ring of nfl + 2 device buffers)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import wfl_asr_amd  # noqa: E402,F401
import synthetic as synth  # noqa: E402
from wfl_asr_amd.tagger import BIOPhonemeTagger  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    cfg = synth.baseline_config(1)
    labels = synth.make_labels(141)
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, len(labels), seed=1).items()}
    model = BIOPhonemeTagger(cfg, labels)
    model.load_state_dict(sd)
    model.to(dev).eval()
    B, L = 16, 480000
    base = synth.make_batch(10000, B, L, seed=1).astype(np.float32)
    wav = torch.from_numpy(base).to(dev)
    lang = (torch.arange(B, device=dev) % cfg["model"]["num_languages"]).to(torch.int32)
    nfl = 2
    T = model.num_frames(L)
    words = B * T * 4 + 1
    host_bufs = [torch.zeros(words, dtype=torch.int32).pin_memory() for _ in range(nfl)]
    streams = [torch.cuda.Stream(dev) for _ in range(nfl)]
    copy_stream = torch.cuda.Stream(dev)
    pinned = [torch.from_numpy(np.roll(base, j, axis=0).copy()).pin_memory() for j in range(8)]
    steps = int(os.environ.get("LAB_STEPS", "40"))

    def run(mode):
        ring = nfl + (2 if mode.endswith("ahead2") else 1)
        dev_in = [torch.empty(B, L, dtype=torch.float32, device=dev) for _ in range(ring)]
        copied = [torch.cuda.Event() for _ in range(ring)]
        consumed = [torch.cuda.Event() for _ in range(ring)]
        for k in range(ring):
            consumed[k].record(streams[0])

        def step(i):
            slot = i % nfl
            k = i % ring
            x = wav
            if mode in ("copy_stream", "copy_stream_ahead2", "copy_unused", "events_only"):
                with torch.cuda.stream(copy_stream):
                    copy_stream.wait_event(consumed[k])
                    if mode != "events_only":
                        dev_in[k].copy_(pinned[i % 8], non_blocking=True)
                    copied[k].record(copy_stream)
                streams[slot].wait_event(copied[k])
                if mode in ("copy_stream", "copy_stream_ahead2"):
                    x = dev_in[k]
            with torch.cuda.stream(streams[slot]):
                if mode == "same_stream":
                    dev_in[k].copy_(pinned[i % 8], non_blocking=True)
                    x = dev_in[k]
                out = model.label(x, lang, threshold=0.5, slot=slot)
                host_bufs[slot].copy_(out.packed, non_blocking=True)
            if mode != "resident" and mode != "same_stream":
                consumed[k].record(streams[slot])

        for i in range(6):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        print(f"  {mode:20s} {1e3 * el / steps:7.3f} ms per step  {B * 30.0 * steps / el:9.0f} audio-s/s", flush=True)

    # the crossing alone: pinned -> device, nothing else on the GPU (boxes differ: 6-8 GB/s on some, 25+ on others)
    dst = torch.empty(B, L, dtype=torch.float32, device=dev)
    for rnd in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(16):
            dst.copy_(pinned[i % 8], non_blocking=True)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        print(f"  copy alone: {16 * B * L * 4 / el / 1e9:6.1f} GB/s ({1e3 * el / 16:.3f} ms per 16 x 30 s batch of float32)", flush=True)
    for rnd in range(2):
        for mode in ("resident", "copy_unused", "events_only", "copy_stream", "copy_stream_ahead2", "same_stream"):
            run(mode)


if __name__ == "__main__":
    main()

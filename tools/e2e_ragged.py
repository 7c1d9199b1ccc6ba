#!/usr/bin/env python3
"""Diagnostic (not product): a folder of WavLM clips of DIFFERENT lengths (2 - 10 s) through Labeler.label_files, with ragged batches
(per-clip lengths inside one forward) and with one length per forward (WFL_RAGGED=0, what rounds 1-2 did: B = 1 for distinct lengths).
usage: e2e_ragged.py [--files 128] [--config-index 0|2]"""
import argparse
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from wfl_asr_amd import audio as A
from wfl_asr_amd import infer as I
import synthetic as synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--files", type=int, default=128)
    ap.add_argument("--config-index", type=int, default=0)
    args = ap.parse_args()
    d = tempfile.mkdtemp(prefix="wfl_ragged_")
    cfg = synth.baseline_config(args.config_index)
    cfg["output"] = {"save_dir": os.path.join(d, "save")}
    cfg["postprocess"] = {"median_filter": 3, "merge_segments": "right", "confidence_threshold": 0.5}
    os.makedirs(cfg["output"]["save_dir"])
    labels = synth.make_labels(70)
    with open(os.path.join(cfg["output"]["save_dir"], "phonemes.txt"), "w") as f:
        f.write("\n".join(labels) + "\n")
    with open(os.path.join(cfg["output"]["save_dir"], "langs.txt"), "w") as f:
        f.write("en,0\nja,1\n")
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, len(labels), seed=1).items()}
    rng = np.random.default_rng(3)
    paths, total = [], 0.0
    os.makedirs(os.path.join(d, "wavs"))
    base = synth.make_clip(7000, 160000, seed=1) * 0.8
    for i in range(args.files):
        n = int(rng.integers(32000, 160000))
        p = os.path.join(d, "wavs", f"{i:04d}.wav")
        A.write_wav(p, np.roll(base, 997 * i)[:n], 16000)
        paths.append(p)
        total += n / 16000
    lab = I.Labeler(cfg, sd, "cuda")
    res = {}
    for mode in ("1", "0"):
        os.environ["WFL_RAGGED"] = mode
        lab.label_files(paths[:8], lang_id=0, confidence_threshold=0.5, verbose=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res[mode] = lab.label_files(paths, lang_id=0, confidence_threshold=0.5, verbose=False)
        dt = time.perf_counter() - t0
        print(f"WFL_RAGGED={mode}: {args.files} files, {total:.0f} audio-s in {dt:.3f} s = {total / dt:.0f} audio-s/s (batch size {lab.batch_size})")
    print("identical segments:", res["1"] == res["0"])


if __name__ == "__main__":
    main()

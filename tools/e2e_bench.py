#!/usr/bin/env python3
"""Diagnostic (not product): end-to-end labeling of a folder of synthetic 30 s WAV files through wfl_asr_amd.infer.Labeler
(BASELINE config 2 model), with a per-phase breakdown: WAV decode + normalise, batched forward (pipelined, two batches in
flight), native segment decode + merge, .lab writing.  Reports files/s and audio-s/s for the whole loop."""
import argparse
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import yaml

from wfl_asr_amd import audio as A
from wfl_asr_amd import infer as I
from wfl_asr_amd import native_post as npost
from wfl_asr_amd import postprocess as pp
import synthetic as synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--files", type=int, default=96)
    ap.add_argument("--full-head", action="store_true", help="the default config.yaml head (BiLSTM + Conformer + dilated) at the "
                    "Labeler's own batch size and batches in flight; prints the end-to-end rate only")
    ap.add_argument("--rate", type=int, default=16000, help="sample rate of the files on disk (other than 16000: every file takes the "
                    "general ingest path -- decode, resample, chunk -- and the end-to-end rate is printed alone)")
    ap.add_argument("--wavlm", action="store_true", help="BASELINE configs[2] (WavLM-large + 2-layer BiLSTM + dilated stack) on files of "
                    "6-10 s: the ragged by-length loop (Labeler._forward_items_by_length); prints the end-to-end rate only")
    args = ap.parse_args()
    d = tempfile.mkdtemp(prefix="wfl_e2e_")
    cfg = synth.baseline_config(2) if args.wavlm else (synth.base_config("whisper") if args.full_head else synth.baseline_config(1))
    cfg["output"] = {"save_dir": os.path.join(d, "save")}
    cfg["postprocess"] = {"median_filter": 3, "merge_segments": "right", "confidence_threshold": 0.5}
    cfg.setdefault("data", {})["sample_rate"] = 16000
    os.makedirs(cfg["output"]["save_dir"])
    labels = synth.make_labels(70)
    with open(os.path.join(cfg["output"]["save_dir"], "phonemes.txt"), "w") as f:
        f.write("\n".join(labels) + "\n")
    with open(os.path.join(cfg["output"]["save_dir"], "langs.txt"), "w") as f:
        f.write("en,0\nja,1\n")
    # (the WavLM-large + BiLSTM stack leaves the synthetic classifier's input ~11 x smaller than Whisper-base's does: at the usual gain no
    #  tag ever crosses the confidence threshold and the decode / merge / .lab legs of the run would have nothing to do)
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, len(labels), seed=1, cls_gain=66.0 if args.wavlm else 6.0).items()}
    wavs = os.path.join(d, "wavs")
    os.makedirs(wavs)
    base = [synth.make_clip(5000 + i, 480000, seed=1) * 0.8 for i in range(8)]
    paths = []
    if args.rate != 16000:
        base = [A.resample(b.astype(np.float64), 16000, args.rate).astype(np.float32) for b in base]
    secs = []
    for i in range(args.files):
        p = os.path.join(wavs, f"{i:04d}.wav")
        n = len(base[i % 8]) if not args.wavlm else int(args.rate * (6.0 + 4.0 * ((i * 37) % 101) / 100.0))     # 6 .. 10 s
        A.write_wav(p, base[i % 8][:n], args.rate)
        secs.append(n / args.rate)
        paths.append(p)
    lab = I.Labeler(cfg, sd, "cuda", batch_size=None if (args.full_head or args.wavlm) else 16)
    if args.wavlm:
        lab.label_files(paths[:2 * lab.batch_size], lang_id=0, confidence_threshold=0.5, verbose=False)      # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = lab.label_files(paths, lang_id=0, confidence_threshold=0.5, verbose=False)
        t_all = time.perf_counter() - t0
        print(f"WavLM-large + BiLSTM + dilated (BASELINE configs[2]) | files {len(paths)} x 6-10 s at {args.rate} Hz | rows per forward {lab.batch_size}, "
              f"batches in flight {lab.n_inflight} | label_files end to end {1e3 * t_all / len(paths):.2f} ms/file = {sum(secs) / t_all:.0f} audio-s/s; "
              f"segments/file {np.mean([len(o) for o in out]):.0f}")
        # the legs one after the other (what the loop above overlaps, or does not)
        from concurrent.futures import ThreadPoolExecutor
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
            items = [c for chunks in ex.map(lambda q: A.load_items(q, 16000), paths) for c in chunks]
        t_load = time.perf_counter() - t0
        t0 = time.perf_counter()
        decided = lab._forward_items(items, 0, 0.5)
        torch.cuda.synchronize()
        t_fwd = time.perf_counter() - t0
        t0 = time.perf_counter()
        segs = [lab._segments_of_item(ids, offs, "en") for ids, offs in decided]
        t_post = time.perf_counter() - t0
        print(f"   legs alone: load (16 threads) {1e3 * t_load:.0f} ms | forward of all {len(items)} items (sorted by length, pipelined) {1e3 * t_fwd:.0f} ms = "
              f"{sum(secs) / t_fwd:.0f} audio-s/s | decode + median + merge {1e3 * t_post:.0f} ms | end to end above {1e3 * t_all:.0f} ms")
        return
    lab.label_files(paths[:4 * lab.batch_size], lang_id=0, confidence_threshold=0.5, verbose=False)      # warm-up
    torch.cuda.synchronize()
    if args.full_head or args.rate != 16000:
        t0 = time.perf_counter()
        out = lab.label_files(paths, lang_id=0, confidence_threshold=0.5, verbose=False)
        t_all = time.perf_counter() - t0
        n = len(paths)
        print(f"{'default config.yaml head' if args.full_head else 'cfg2 model'} | files {n} x 30 s at {args.rate} Hz | rows per forward {lab.batch_size}, batches in flight {lab.n_inflight} | "
              f"label_files end to end {1e3 * t_all / n:.2f} ms/file = {30 * n / t_all:.0f} audio-s/s; segments/file {np.mean([len(o) for o in out]):.0f}")
        return

    t0 = time.perf_counter()
    clips = [A.load_clip(p, 16000) for p in paths]
    t_load = time.perf_counter() - t0
    items = [A.chunk_clip(c, 16000)[0] for c in clips]
    t0 = time.perf_counter()
    decided = lab._forward_items(items, 0, 0.5)
    torch.cuda.synchronize()
    t_fwd = time.perf_counter() - t0
    t0 = time.perf_counter()
    segs = [lab._segments_of_item(ids, offs, "en") for ids, offs in decided]
    names = lab._names_for("en")[1]
    merged = [npost.merge_segments(s, e, ph, "right") for s, e, ph in segs]
    t_post = time.perf_counter() - t0
    t0 = time.perf_counter()
    for (ids, offs) in decided:
        tags = [lab.model.id2label[int(i)] for i in pp.median_filter_ids(ids, 3)]
        pp.merge_adjacent_segments(pp.decode_bio_tags(tags, offsets=offs), "right")
    t_post_py = time.perf_counter() - t0
    t0 = time.perf_counter()
    for p, (s, e, ph) in zip(paths, merged):
        with open(p[:-4] + ".lab", "wb") as f:
            f.write(npost.format_lab(s, e, ph, names))
    t_write = time.perf_counter() - t0
    t0 = time.perf_counter()
    out = lab.label_files(paths, lang_id=0, confidence_threshold=0.5, verbose=False)
    t_all = time.perf_counter() - t0
    n = len(paths)
    print(f"files {n} x 30 s | load {1e3 * t_load / n:.2f} ms/file | forward (pipelined, incl. pinned fill + H2D/D2H) "
          f"{1e3 * t_fwd / n:.3f} ms/file = {30 * n / t_fwd:.0f} audio-s/s | native decode+median+merge {1e6 * t_post / n:.0f} us/file "
          f"(python reference logic {1e6 * t_post_py / n:.0f} us/file, x{t_post_py / max(t_post, 1e-9):.0f}) | .lab write {1e6 * t_write / n:.0f} us/file | "
          f"label_files end to end {1e3 * t_all / n:.2f} ms/file = {30 * n / t_all:.0f} audio-s/s; segments/file {np.mean([len(o) for o in out]):.0f}")


if __name__ == "__main__":
    main()

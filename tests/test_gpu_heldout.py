"""-m gpu: BIO-tag index parity on a HELD-OUT synthetic set (BASELINE north_star): 64 clips whose generator seeds
are disjoint from every fixture, tuning run and other test (seed 777, clip indices 50000..50063), lengths mixed 1-30 s, labelled by
the product loop (`Labeler`, ragged `lens`, batches of 16, two batches in flight) and by the oracle on the box, one clip per
oracle call exactly like the reference's loop (/root/reference/infer.py:237-307: per segment peak normalisation, forward,
suppress_low_confidence at the config's threshold, label ids, decode_bio_tags).

Reported (gpurun_out/parity_stats.jsonl) and asserted, against both targets of tests/test_gpu_model.py:
  raw id mismatch rate over ALL frames, graded fraction (margin > tau and |max-prob - thr| > band), mismatches on graded frames
  (must be 0), and the `.lab` level: clips whose segment label sequence is identical, and on those the largest boundary shift.
Measured on MI355X (round 2): target A raw 1.98 %, graded 61.6 %; target B raw 1.49 %, graded 79.2 %; 0 graded mismatches in
96 000 frames either way.  (The set is harder than the cfg2 fixture: two thirds of its frames are zero padding behind short
clips, where the synthetic model's logits sit close together.)"""
import os

import numpy as np
import pytest
import torch

from oracle import wfl_oracle as O
from wfl_asr_amd import infer as I
from wfl_asr_amd import postprocess as pp
import synthetic as synth
from wfl_asr_amd.archs import resolve_encoder_arch
from test_gpu_model import TAU, BAND, TAU_W, BAND_W, _note

pytestmark = pytest.mark.gpu

N_CLIPS, SEED, CLIP0, THR = 64, 777, 50000, 0.5


def _segments(ids, offsets, labels):
    tags = [labels[int(i)] for i in ids]
    return pp.merge_adjacent_segments(pp.decode_bio_tags(tags, offsets=offsets), "right")


@pytest.mark.parametrize("target,head,n_clips", [("fp32_weights", "cfg2", N_CLIPS), ("bf16_weights", "cfg2", N_CLIPS),
                                                 ("bf16_weights", "default_head", 8), ("fp32_weights_precision_high", "cfg2", N_CLIPS),
                                                 ("fp32_weights_precision_high", "default_head", 8)])
def test_held_out_set_tag_index_parity(target, head, n_clips, tmp_path):
    """head = cfg2: BASELINE configs[1] (Whisper-base + 2 Conformer), all 64 clips.  head = default_head: the reference's default
    config.yaml head (2-layer BiLSTM + 2 Conformer + 2 dilated convs) on the first 8 clips of the same set -- the oracle's BiLSTM is
    a Python time loop, 8 clips keep it under a minute."""
    cfg = synth.baseline_config(1) if head == "cfg2" else synth.base_config("whisper")
    high = target.endswith("precision_high")
    if high:
        # `model.precision: high` (round 3): every GEMM as three bf16 passes over split operands, fp32 sums, hi + lo activations --
        # (the attention too: q, k, v and P as bf16 pairs) -- the switch for callers who need the reference's `.lab`; held to the
        # reference on the checkpoint as given at an eightieth of the default build's tau
        cfg["model"]["precision"] = "high"
    cfg["output"]["save_dir"] = str(tmp_path)
    cfg["postprocess"] = {"median_filter": 1, "merge_segments": "right", "confidence_threshold": THR}
    labels = synth.make_labels(70)
    (tmp_path / "phonemes.txt").write_text("\n".join(labels) + "\n")
    (tmp_path / "langs.txt").write_text("en,0\nja,1\n")
    sd_np = synth.make_state_dict(cfg, len(labels), seed=SEED)
    if target == "bf16_weights":
        sd_np = synth.round_weights_bf16(sd_np)
    tau, band = (TAU_W, BAND_W) if target == "bf16_weights" else ((0.005, 0.002) if high else (TAU, BAND))
    # mixed lengths 1-30 s (a few exactly 30 s, a few very short), peak-normalised like infer.py:235 by the generator
    u = synth.uniform01("heldout.len", N_CLIPS, SEED)[:n_clips]
    secs = np.where(u < 0.1, 30.0, np.where(u > 0.9, 1.0 + 2.0 * u, 1.0 + 29.0 * u))
    clips = [synth.make_clip(CLIP0 + i, int(round(float(s) * 16000)), seed=SEED) for i, s in enumerate(secs)]
    lang_id = 1
    lab = I.Labeler(cfg, {k: torch.from_numpy(v) for k, v in sd_np.items()}, device="cuda", batch_size=16)
    got = lab._forward_items(clips, lang_id, THR)                       # [(ids [T], offsets [T, 2])] per clip
    lab.model.check(16, 480000, slot=0)
    lab.model.check(16, 480000, slot=1)

    enc, arch = resolve_encoder_arch(cfg["model"])
    sd = O.to_torch_state_dict(sd_np)
    hc = synth.head_config(cfg["model"])
    o_id = labels.index("O")
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    raw_bad = graded = graded_bad = frames = same_seq = n_seg = 0
    worst = shift = 0.0
    for i, x in enumerate(clips):
        lg, of = O.forward(torch.from_numpy(x)[None], torch.tensor([lang_id]), sd, enc, arch, hc)     # B = 1, like infer.py:261
        ids_ref, maxp, arg, margin = O.tags_from_logits(lg[0], o_id, THR)
        ids, offs = got[i]
        ids_t = torch.from_numpy(ids.astype(np.int64))
        safe = (margin > tau) & ((maxp - THR).abs() > band)
        raw_bad += int((ids_t != ids_ref).sum())
        graded += int(safe.sum())
        graded_bad += int((ids_t != ids_ref)[safe].sum())
        frames += ids_ref.numel()
        worst = max(worst, float((torch.from_numpy(offs) - of[0]).abs().max()))
        a, b = _segments(ids, offs, labels), _segments(ids_ref.numpy(), of[0].numpy(), labels)
        n_seg += len(b)
        if [ph for _, _, ph in a] == [ph for _, _, ph in b]:
            same_seq += 1
            for (s0, e0, _), (s1, e1, _) in zip(a, b):
                shift = max(shift, abs(s0 - s1), abs(e0 - e1))
    _note("heldout_" + head + "_" + target, clips=n_clips, frames=frames, audio_s=float(secs.sum()), tau=tau, band=band,
          raw_mismatch_rate=raw_bad / frames, graded_frac=graded / frames, graded_mismatches=graded_bad, offsets_max=worst,
          clips_with_identical_label_sequence=same_seq, reference_segments=n_seg, max_boundary_shift_s=shift)
    assert graded_bad == 0
    assert worst <= 0.02
    if high:
        # measured: 3 of 96 000 raw decisions differ (default build: 1.98 %); 62 of 64 clips give the reference's label sequence (default:
        # 5) with boundaries within 50 us; offsets within 0.003
        assert raw_bad / frames < 0.0005 and graded / frames >= 0.97 and same_seq >= (58 * n_clips) // 64, (raw_bad / frames, graded / frames, same_seq)
        assert worst <= 0.004
    elif head != "cfg2":
        assert graded / frames >= 0.50 and raw_bad / frames < 0.04
    elif target == "bf16_weights":
        assert graded / frames >= 0.75 and raw_bad / frames < 0.02
    else:
        assert graded / frames >= 0.58 and raw_bad / frames < 0.025

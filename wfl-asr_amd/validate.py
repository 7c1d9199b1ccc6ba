"""Validation pass on the fast path: what a `train.py` maintainer calls instead of the reference's
`evaluate` so that training-time validation runs through the HIP forward.

  evaluate                  /root/reference/train.py:456-545  (same arguments, same scalars, same return value; no figures)
  framewise_accuracy        /root/reference/train.py:98-102
  phoneme_error_rate        /root/reference/train.py:104-125   (edit distance of the phoneme sequences / reference length)
  timing_error              /root/reference/train.py:127-147   (first predicted segment with the ground-truth segment's phoneme)
  phoneme_name              /root/reference/train.py:89-96

The forward is `model(input_values, lang_ids, max_label_len=...)` (tagger.py: `wfl_encode` -> pad / truncate -> `wfl_head`), the decode is
the fixture-pinned host logic of postprocess.py.  The metrics are a few lines of host arithmetic each and are PINNED since round 4:
tests/golden/metrics.json holds the outputs of the reference's own four functions on 240 seeded segment pairs and 40 logits / label
tensors (tests/golden/make_golden.py imports train.py with empty stubs for the modules this container lacks); tests/test_validate.py
holds these functions, and the loop-for-loop restatement in oracle/wfl_metrics.py, to them.
"""
from __future__ import annotations

import numpy as np
import torch

from .postprocess import decode_bio_tags, median_filter_ids, merge_adjacent_segments


def phoneme_name(seg):
    """The bare phoneme of a segment or label: the third field of a (start, end, ph) triple, unwrapped from one-element containers,
    without a "lang/" prefix (train.py:89-96)."""
    ph = seg[2] if isinstance(seg, (tuple, list)) and len(seg) == 3 else seg
    while isinstance(ph, (tuple, list)) and len(ph) == 1:
        ph = ph[0]
    return str(ph).split("/")[-1]


def framewise_accuracy(pred_ids, labels):
    """Share of frames whose predicted id equals the label (every frame counts, the padding label included, as in the reference)."""
    pred_ids = torch.as_tensor(pred_ids)
    labels = torch.as_tensor(labels).to(pred_ids.device)
    n = labels.numel()
    return float((pred_ids == labels).sum().item()) / n if n > 0 else 0.0


def phoneme_error_rate(pred_segments, gt_segments):
    """Levenshtein distance between the two phoneme sequences (names compared as they are) over max(len(gt), 1)."""
    pred = [ph for _, _, ph in pred_segments]
    gt = [ph for _, _, ph in gt_segments]
    prev = np.arange(len(pred) + 1)
    for i, g in enumerate(gt, 1):
        cur = np.empty_like(prev)
        cur[0] = i
        sub = prev[:-1] + np.fromiter((0 if g == p else 1 for p in pred), dtype=prev.dtype, count=len(pred))
        best = np.minimum(sub, prev[1:] + 1)                   # substitution / match, deletion
        # insertions chain along the row: cur[j] = min(best[j-1], cur[j-1] + 1)
        run = cur[0]
        for j in range(1, len(pred) + 1):
            run = min(best[j - 1], run + 1)
            cur[j] = run
        prev = cur
    return float(prev[-1]) / max(len(gt), 1)


def timing_error(pred_segments, gt_segments):
    """For every ground-truth segment, the FIRST predicted segment with the same bare phoneme: mean of (|start error| + |end error|) / 2
    over the matches, divided by the mean duration of the matched ground-truth segments; 0 without a match."""
    first = {}
    for s, e, ph in pred_segments:
        first.setdefault(phoneme_name(ph), (s, e))
    errs, durs = [], []
    for s, e, ph in gt_segments:
        hit = first.get(phoneme_name(ph))
        if hit is not None:
            errs.append(abs(s - hit[0]) + abs(e - hit[1]))
            durs.append(e - s)
    if not errs:
        return 0.0
    mean_dur = float(np.mean(durs))
    return float(np.mean(errs)) / 2 / mean_dur if mean_dur > 0 else 0.0


def evaluate(model, val_loader, label_list, config, writer=None, step=0, criterion=None, id2lang=None, merge_map=None):
    """One pass over `val_loader` (batches of train.py's collate: input_values, label_ids, wavs, ground-truth segments, paths, lang_ids,
    label_lengths): loss (when a criterion is given), frame accuracy, phoneme error rate and timing error per clip, averaged; the four
    scalars go to `writer.add_scalar("val/...")` when a writer is given, the summary line is printed, the mean loss is returned."""
    if hasattr(model, "eval"):
        model.eval()
    dev = getattr(model, "device", None)
    frame_duration = config["data"].get("frame_duration", 0.02)      # (evaluate, unlike infer.py, honours the config value)
    median = config["postprocess"]["median_filter"]
    merge = config["postprocess"]["merge_segments"]
    id2label = dict(enumerate(label_list))
    losses, acc, per, ter, count = [], 0.0, 0.0, 0.0, 0
    with torch.no_grad():
        for batch in val_loader:
            input_values, label_ids, _wavs, segments_gt_batch, _paths, lang_ids, label_lengths = batch
            if dev is not None:
                input_values, label_ids, lang_ids = input_values.to(dev), label_ids.to(dev), lang_ids.to(dev)
            max_label_len = int(torch.max(label_lengths)) if label_lengths.numel() > 0 else 0
            logits, offsets = model(input_values, lang_ids, max_label_len=max_label_len)
            if criterion is not None:
                losses.append(float(criterion(logits.reshape(-1, logits.size(-1)), label_ids.reshape(-1)).item()))
            pred_dev = torch.argmax(logits, dim=-1)
            pred_all = pred_dev.cpu().numpy()
            offs_all = offsets.cpu().numpy() if offsets is not None else None
            for j in range(input_values.size(0)):
                n = int(label_lengths[j])
                ids = pred_all[j, :n]
                if median > 1:
                    ids = median_filter_ids(ids, median)
                tags = [id2label[int(i)] for i in ids]
                segs = decode_bio_tags(tags, frame_duration=frame_duration, offsets=offs_all[j, :n] if offs_all is not None else None)
                if merge != "none":
                    segs = merge_adjacent_segments(segs, mode=merge)
                gt = segments_gt_batch[j]
                if isinstance(gt, list) and len(gt) == 1 and isinstance(gt[0], list):
                    gt = gt[0]
                acc += framewise_accuracy(pred_dev[j, :n], label_ids[j, :n])      # (the unfiltered argmax, as in the reference)
                per += phoneme_error_rate(segs, gt)
                ter += timing_error(segs, gt)
                count += 1
    avg_loss = sum(losses) / len(losses) if losses else 0
    avg_acc, avg_per, avg_ter = (acc / count, per / count, ter / count) if count else (0, 0, 0)
    if writer is not None:
        writer.add_scalar("val/loss", avg_loss, step)
        writer.add_scalar("val/accuracy", avg_acc, step)
        writer.add_scalar("val/per", avg_per, step)
        writer.add_scalar("val/ter", avg_ter, step)
    print(f"\n[Validation] Loss: {avg_loss:.4f} | Acc: {avg_acc*100:.2f}% | PER: {avg_per:.3f} | TER: {avg_ter:.3f}")
    evaluate.last = {"loss": avg_loss, "accuracy": avg_acc, "per": avg_per, "ter": avg_ter, "clips": count}
    return avg_loss

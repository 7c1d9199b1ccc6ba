"""ORACLE — test infrastructure, not product code.

Plain-loop restatements of the reference's validation metrics, used ONLY by tests/test_validate.py as the checker of
wfl-asr_amd/validate.py.  Nothing under `wfl-asr_amd/` may import this file.

PARITY UNPINNED: the reference holds no fixtures for these functions and its train.py cannot be imported in the build container
(it needs tensorboard, which is absent), so no vectors could be generated from it; each function restates the cited lines.
  ref = /root/reference/train.py
"""
from __future__ import annotations


def bare_name(seg):
    """ref:89-96"""
    ph = seg[2] if isinstance(seg, (tuple, list)) and len(seg) == 3 else seg
    while isinstance(ph, (tuple, list)) and len(ph) == 1:
        ph = ph[0]
    return str(ph).split("/")[-1]


def edit_rate(pred_segments, gt_segments):
    """ref:104-125: full-table Levenshtein distance between the phoneme sequences / max(len(gt), 1)"""
    hyp = [s[2] for s in pred_segments]
    ref = [s[2] for s in gt_segments]
    table = [[0] * (len(hyp) + 1) for _ in range(len(ref) + 1)]
    for r in range(len(ref) + 1):
        table[r][0] = r
    for c in range(len(hyp) + 1):
        table[0][c] = c
    for r in range(1, len(ref) + 1):
        for c in range(1, len(hyp) + 1):
            same = ref[r - 1] == hyp[c - 1]
            table[r][c] = min(table[r - 1][c] + 1, table[r][c - 1] + 1, table[r - 1][c - 1] + (0 if same else 1))
    return table[len(ref)][len(hyp)] / max(len(ref), 1)


def timing_rate(pred_segments, gt_segments):
    """ref:127-147: per ground-truth segment the first prediction of the same bare phoneme; mean(start + end error) / 2 / mean duration"""
    sums, spans = [], []
    for g0, g1, gph in gt_segments:
        for p0, p1, pph in pred_segments:
            if bare_name(pph) == bare_name(gph):
                sums.append(abs(g0 - p0) + abs(g1 - p1))
                spans.append(g1 - g0)
                break
    if not sums:
        return 0.0
    mean_span = sum(spans) / len(spans)
    return (sum(sums) / len(sums)) / 2 / mean_span if mean_span > 0 else 0.0


def frame_accuracy(logits, labels):
    """ref:98-102 (logits [.., C], labels [..]: nested lists or arrays)"""
    import numpy as np
    lg, lb = np.asarray(logits), np.asarray(labels)
    return float((lg.argmax(-1) == lb).sum()) / lb.size if lb.size else 0.0

"""Host-side label logic: BIO tags -> segments -> `.lab` (L1 of SURVEY.md §1).

O(frames) work on tiny data that stays on the host; semantics (including the idiosyncratic
ones) follow the reference exactly and are pinned by fixtures generated from the reference's
own functions (tests/golden/postprocess.json):

  decode_bio_tags           /root/reference/utils.py:10-74
  save_lab                  /root/reference/utils.py:76-81
  load_phoneme_list         /root/reference/utils.py:83-85
  merge_adjacent_segments   /root/reference/utils.py:148-186
  load_langs / merge map    /root/reference/utils.py:188-211
  align_phoneme_list        /root/reference/infer.py:30-60
  split_audio               /root/reference/infer.py:19-28
  suppress_low_confidence   /root/reference/infer.py:86-96
  median filter             scipy.ndimage.median_filter(ids, size=k) as called at infer.py:170-171, 298-299
"""
from __future__ import annotations

import json
import os

import numpy as np

HTK_TIME_FACTOR = 1e7      # .lab times are in 100 ns units
FRAME_DURATION = 0.02      # infer.py:12 hard-codes 20 ms frames (config data.frame_duration is ignored)
MAX_SEGMENT_DURATION = 30.0


def _offset(offsets, idx, col):
    v = offsets[idx][col]
    return float(v.item()) if hasattr(v, "item") else float(v)


def decode_bio_tags(tags, frame_duration=FRAME_DURATION, offsets=None):
    """tags (list[str]) + optional per-frame (start, end) sub-frame offsets -> [(start_s, end_s, ph)].

    A run closes at "O", at any "B-", or at an "I-x" whose x differs from the open phoneme (which
    then *opens* a run for x).  The closing index is the index of the frame that closed the run,
    or len-1 at end of input; times are (idx + offset) * frame_duration, offset 0.5 without offsets.
    """
    segments = []
    open_ph = None
    open_idx = None
    n_off = len(offsets) if offsets is not None else 0

    def close(end_idx, at_eof=False):
        use_off = offsets is not None and (not at_eof or (open_idx < n_off and end_idx < n_off))
        if use_off:
            s = (open_idx + _offset(offsets, open_idx, 0)) * frame_duration
            e = (end_idx + _offset(offsets, end_idx, 1)) * frame_duration
        else:
            s = (open_idx + 0.5) * frame_duration
            e = (end_idx + 0.5) * frame_duration
        segments.append((s, e, open_ph))

    for i, tag in enumerate(tags):
        if tag == "O":
            if open_ph is not None:
                close(i)
                open_ph, open_idx = None, None
        elif tag.startswith("B-"):
            if open_ph is not None:
                close(i)
            open_ph, open_idx = tag[2:], i
        elif tag.startswith("I-"):
            ph = tag[2:]
            if ph != open_ph:
                if open_ph is not None:
                    close(i)
                open_ph, open_idx = ph, i
    if open_ph is not None:
        close(len(tags) - 1, at_eof=True)
    return segments


def save_lab(path, segments):
    """HTK label file; times truncated (not rounded) to 100 ns integers."""
    with open(path, "w", encoding="utf-8") as f:
        for start, end, ph in segments:
            f.write(f"{int(start * HTK_TIME_FACTOR)} {int(end * HTK_TIME_FACTOR)} {ph}\n")


def load_phoneme_list(path):
    with open(path, "r", encoding="utf-8") as f:
        return [ln.strip() for ln in f if ln.strip()]


def load_langs(lang_path):
    lang2id = {}
    with open(lang_path, "r", encoding="utf-8") as f:
        for line in f:
            lang, idx = line.strip().split(",")
            lang2id[lang] = int(idx)
    return lang2id


def load_phoneme_merge_map(path):
    if not os.path.exists(path):
        return None
    with open(path, "r", encoding="utf-8") as f:
        return json.load(f)


def canonical_to_lang(phoneme, lang, merge_map):
    if not merge_map:
        return phoneme
    entry = merge_map.get(phoneme)
    if entry is None:
        return phoneme
    return entry.get(lang, phoneme)


def merge_adjacent_segments(segments, mode="right"):
    """Merge equal-label neighbours.  `right` and `left` both extend the earlier segment's end;
    `previous` reproduces the reference's own rule (utils.py:170-183): from the third segment on,
    a segment equal to its predecessor collapses the last two merged entries into the one before."""
    if not segments or mode == "none":
        return segments
    if mode in ("right", "left"):
        merged = []
        for i, seg in enumerate(segments):
            if i > 0 and seg[2] == segments[i - 1][2] and merged:
                s0, _, ph = merged.pop()
                merged.append((s0, seg[1], ph))
            else:
                merged.append(seg)
        return merged
    if mode == "previous":
        merged = []
        for i, seg in enumerate(segments):
            if i > 1 and segments[i - 1][2] == seg[2]:
                if len(merged) >= 2:
                    p0 = merged[-2]
                    merged.pop()
                    merged[-1] = (p0[0], seg[1], p0[2])
                else:
                    merged.append(seg)
            else:
                merged.append(seg)
        return merged
    raise ValueError(f"Unsupported merge mode: {mode}")


def align_phoneme_list(segments_pred, forced_list):
    """Greedy in-order match of a forced phoneme list onto predicted segments; unmatched forced
    entries take the next unused predicted segment; output is in forced-list order."""
    n_pred = len(segments_pred)
    used = set()
    assigned = [None] * len(forced_list)
    cursor = 0
    for fi, ph in enumerate(forced_list):
        for pi in range(cursor, n_pred):
            if pi not in used and segments_pred[pi][2] == ph:
                assigned[fi] = pi
                used.add(pi)
                cursor = pi + 1
                break
    spare = 0
    for fi in range(len(forced_list)):
        if assigned[fi] is None:
            while spare < n_pred and spare in used:
                spare += 1
            if spare < n_pred:
                assigned[fi] = spare
                used.add(spare)
                spare += 1
    out = []
    for fi, ph in enumerate(forced_list):
        pi = assigned[fi]
        if pi is not None and pi < n_pred:
            s, e, _ = segments_pred[pi]
            out.append((s, e, ph))
    return out


def split_audio(audio, sr, max_duration=MAX_SEGMENT_DURATION):
    """Non-overlapping chunks of int(max_duration * sr) samples (last one shorter)."""
    step = int(max_duration * sr)
    return [audio[s:min(s + step, len(audio))] for s in range(0, len(audio), step)]


def median_filter_ids(ids, size):
    """scipy.ndimage.median_filter(ids, size=size) for 1-D integer ids: window offsets
    [-(size//2), size - size//2 - 1], half-sample-symmetric ("reflect") boundary, rank size//2."""
    a = np.asarray(ids)
    if size <= 1 or a.size == 0:
        return a.copy()
    left = size // 2
    right = size - left - 1
    n = a.size
    # numpy 'symmetric' padding == scipy 'reflect'; repeat reflection for windows longer than the input
    idx = np.arange(-left, n + right)
    period = 2 * n
    idx = np.mod(idx, period)
    idx = np.where(idx >= n, period - 1 - idx, idx)
    padded = a[idx]
    win = np.lib.stride_tricks.sliding_window_view(padded, size)
    return np.sort(win, axis=1)[:, size // 2].astype(a.dtype)


def softmax_max(logits):
    """float32 softmax over the last axis -> (max prob, argmax); same arithmetic order as
    torch.softmax on CPU is not guaranteed bit-for-bit, only to ~1e-7 (used for host fallbacks of
    *postprocessing inputs* such as cached/averaged logits, never for the model forward)."""
    x = np.asarray(logits, dtype=np.float32)
    m = x.max(axis=-1, keepdims=True)
    e = np.exp(x - m)
    p = e / e.sum(axis=-1, keepdims=True)
    return p.max(axis=-1), p.argmax(axis=-1)


def tags_from_decisions(argmax_ids, max_probs, id2label, threshold):
    """The decision rule of suppress_low_confidence given (argmax, max prob) from the GPU."""
    return ["O" if p < threshold else id2label[int(i)] for p, i in zip(max_probs, argmax_ids)]


def suppress_low_confidence(logits, id2label, threshold=0.5):
    """Reference signature (infer.py:86-96): [T, C] logits -> list of tag strings."""
    if hasattr(logits, "detach"):
        logits = logits.detach().cpu().numpy()
    maxp, arg = softmax_max(logits)
    return tags_from_decisions(arg, maxp, id2label, threshold)

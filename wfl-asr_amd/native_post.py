"""Native (C ABI, host-only) label logic: ids + offsets of a clip -> segments -> `.lab` text.

Thin ctypes layer over `wfl_host_*` (csrc/hostpost.hip).  Same semantics, bit for bit, as postprocess.py's
`median_filter_ids`, `decode_bio_tags`, `merge_adjacent_segments` and `save_lab` (which are pinned to the reference's own
functions by tests/golden/postprocess.json); tests/test_native_post.py holds the two implementations against each other.
It exists because at MI355X labeling speed the per-frame Python of /root/reference/utils.py:10-74 is the bottleneck.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib

MERGE_MODES = {"none": 0, "right": 1, "left": 2, "previous": 3}


class LabelTable:
    """Per-label lookup built once from the tag strings (phonemes.txt): kind (O / B- / I- / other) and phoneme index."""

    def __init__(self, label_list):
        self.names = []                       # phoneme strings, index = phoneme id
        index = {}
        kind = np.full(len(label_list), 3, np.int32)
        phon = np.full(len(label_list), -1, np.int32)
        for i, tag in enumerate(label_list):
            if tag == "O":
                kind[i] = 0
            elif tag.startswith("B-") or tag.startswith("I-"):
                kind[i] = 1 if tag[0] == "B" else 2
                ph = tag[2:]
                if ph not in index:
                    index[ph] = len(self.names)
                    self.names.append(ph)
                phon[i] = index[ph]
        self.kind, self.phon = kind, phon
        self.n_labels = len(label_list)

    def c_names(self, names=None):
        names = self.names if names is None else names
        enc = [n.encode("utf-8") for n in names]
        arr = (C.c_char_p * max(len(enc), 1))(*enc)
        return arr, enc                        # keep `enc` alive with the array


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def median_filter_ids(ids, size):
    a = np.ascontiguousarray(ids, dtype=np.int32)
    out = np.empty_like(a)
    _lib.check(_lib.load().wfl_host_median_filter(_p(a), a.size, int(size), _p(out)), "wfl_host_median_filter")
    return out


def decode_bio_ids(ids, table: LabelTable, frame_duration=0.02, offsets=None, median=0):
    """-> (start[n] f64, end[n] f64, ph[n] i32) for one clip; `offsets` is [T, 2] float32 or None."""
    lib = _lib.load()
    a = np.ascontiguousarray(ids, dtype=np.int32)
    if median and median > 1:
        a = median_filter_ids(a, median)
    T = a.size
    cap = T + 1
    s = np.empty(cap, np.float64)
    e = np.empty(cap, np.float64)
    ph = np.empty(cap, np.int32)
    off = None
    n_off = 0
    if offsets is not None:
        off = np.ascontiguousarray(offsets, dtype=np.float32)
        n_off = off.shape[0]
    n = lib.wfl_host_decode_bio(_p(a), T, _p(off) if off is not None else None, n_off, _p(table.kind), _p(table.phon),
                                table.n_labels, float(frame_duration), _p(s), _p(e), _p(ph), cap)
    if n < 0:
        raise _lib.WflError(f"wfl_host_decode_bio failed ({n})")
    return s[:n], e[:n], ph[:n]


def merge_segments(s, e, ph, mode):
    """In place on copies; returns the merged (start, end, ph) arrays."""
    s, e, ph = np.array(s, np.float64), np.array(e, np.float64), np.array(ph, np.int32)
    n = _lib.load().wfl_host_merge_segments(_p(s), _p(e), _p(ph), s.size, MERGE_MODES[mode] if isinstance(mode, str) else int(mode))
    if n < 0:
        raise _lib.WflError(f"wfl_host_merge_segments failed ({n})")
    return s[:n], e[:n], ph[:n]


def format_lab(s, e, ph, names) -> bytes:
    """The text save_lab would write (utils.py:76-81)."""
    lib = _lib.load()
    s = np.ascontiguousarray(s, np.float64)
    e = np.ascontiguousarray(e, np.float64)
    ph = np.ascontiguousarray(ph, np.int32)
    enc = [n.encode("utf-8") for n in names]
    arr = (C.c_char_p * max(len(enc), 1))(*enc)
    cap = 48 * max(s.size, 1) + sum(len(enc[i]) for i in ph) + 16
    buf = C.create_string_buffer(cap)
    n = lib.wfl_host_format_lab(_p(s), _p(e), _p(ph), s.size, arr, len(enc), buf, cap)
    if n < 0 or n > cap:
        raise _lib.WflError(f"wfl_host_format_lab failed ({n})")
    return buf.raw[:n]


def format_lab_tuples(segments) -> bytes:
    """[(start_s, end_s, phoneme)] -> the `.lab` text (UTF-8), through the native formatter."""
    names, index = [], {}
    ph = np.empty(len(segments), np.int32)
    for i, (_, _, name) in enumerate(segments):
        if name not in index:
            index[name] = len(names)
            names.append(name)
        ph[i] = index[name]
    s = np.array([a for a, _, _ in segments], np.float64)
    e = np.array([b for _, b, _ in segments], np.float64)
    return format_lab(s, e, ph, names) if len(segments) else b""


def to_tuples(s, e, ph, names):
    """[(start_s, end_s, phoneme)] with Python floats, the reference's return type."""
    return [(float(a), float(b), names[int(p)]) for a, b, p in zip(s, e, ph)]

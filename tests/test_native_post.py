"""Native host label logic (csrc/hostpost.hip through the C ABI) against postprocess.py, which tests/test_postprocess.py
pins to the reference's own functions (tests/golden/postprocess.json).  Host-only code: runs without a GPU."""
import json
import os

import numpy as np
import pytest

from wfl_asr_amd import native_post as N
from wfl_asr_amd import postprocess as pp

GOLD = os.path.join(os.path.dirname(__file__), "golden", "postprocess.json")


def _labels(n_ph=9):
    phs = [f"p{i}" for i in range(n_ph)] + ["SP", "é"]
    return sorted(["O"] + [f"B-{p}" for p in phs] + [f"I-{p}" for p in phs] + ["<unk>"])


@pytest.mark.parametrize("size", [1, 2, 3, 4, 5, 7, 11])
@pytest.mark.parametrize("n", [1, 2, 3, 11, 200])
def test_median_filter_matches_python(size, n):
    rng = np.random.default_rng(100 * size + n)
    ids = rng.integers(0, 9, n).astype(np.int32)
    assert np.array_equal(N.median_filter_ids(ids, size), pp.median_filter_ids(ids, size))


def test_median_filter_known_answers():
    x = [3, 3, 7, 3, 3, 9, 9, 1, 9, 9, 0]                       # SURVEY.md 8c, from scipy itself
    for size, want in ((2, "3 3 7 7 3 9 9 9 9 9 9"), (3, "3 3 3 3 3 9 9 9 9 9 0"), (4, "3 3 3 3 7 9 9 9 9 9 9"),
                       (5, "3 3 3 3 7 3 9 9 9 1 9")):
        assert list(N.median_filter_ids(x, size)) == [int(v) for v in want.split()]


@pytest.mark.parametrize("seed", range(12))
@pytest.mark.parametrize("with_off", [False, True])
def test_decode_merge_format_match_python(seed, with_off, tmp_path):
    labels = _labels()
    table = N.LabelTable(labels)
    rng = np.random.default_rng(seed)
    T = int(rng.integers(1, 400))
    # runs of equal tags, like real output; plus isolated junk
    ids = np.repeat(rng.integers(0, len(labels), T // 3 + 1), rng.integers(1, 6, T // 3 + 1))[:T].astype(np.int32)
    T = ids.size
    offs = rng.random((T, 2)).astype(np.float32) if with_off else None
    if with_off and seed % 3 == 0:
        offs = offs[:max(T - 2, 1)]                              # offsets shorter than the tags (reference guards the EOF close)
    tags = [labels[i] for i in ids]
    try:
        want = pp.decode_bio_tags(tags, frame_duration=0.02, offsets=offs)
    except IndexError:                                           # a run closed beyond the offsets rows: the reference raises too
        with pytest.raises(Exception):
            N.decode_bio_ids(ids, table, 0.02, offs)
        return
    s, e, ph = N.decode_bio_ids(ids, table, 0.02, offs)
    got = N.to_tuples(s, e, ph, table.names)
    assert got == want                                           # doubles compared exactly
    for mode in ("none", "right", "left", "previous"):
        wm = pp.merge_adjacent_segments(list(want), mode=mode)
        ms, me, mp = N.merge_segments(s, e, ph, mode)
        assert N.to_tuples(ms, me, mp, table.names) == wm, mode
        path = tmp_path / f"{mode}.lab"
        pp.save_lab(str(path), wm)
        assert N.format_lab(ms, me, mp, table.names) == path.read_bytes()


def test_known_answer_vectors():
    """outputs captured from the reference's own decode_bio_tags / merge / save_lab."""
    labels = sorted(["O"] + [f"{k}-{p}" for k in "BI" for p in "abcd"] + ["B-SP", "I-SP"])
    table = N.LabelTable(labels)
    tags = "O B-a I-a I-a B-b I-b O O I-c I-c B-a I-d I-d".split()
    ids = np.array([labels.index(t) for t in tags], np.int32)
    s, e, ph = N.decode_bio_ids(ids, table)
    got = [(round(a, 6), round(b, 6), p) for a, b, p in N.to_tuples(s, e, ph, table.names)]
    assert got == [(0.03, 0.09, "a"), (0.09, 0.13, "b"), (0.17, 0.21, "c"), (0.21, 0.23, "a"), (0.23, 0.25, "d")]
    offs = np.linspace(0, 1, 26).reshape(13, 2).astype(np.float32)
    s, e, ph = N.decode_bio_ids(ids, table, 0.02, offs)
    got = [(round(a, 4), round(b, 4), p) for a, b, p in N.to_tuples(s, e, ph, table.names)]
    assert got == [(0.0216, 0.0872, "a"), (0.0864, 0.1304, "b"), (0.1728, 0.2168, "c"), (0.216, 0.2384, "a"), (0.2376, 0.26, "d")]
    names = ["a", "b"]
    s = np.arange(6) * 0.1
    e = s + 0.1
    ph = np.array([0, 0, 1, 1, 1, 0], np.int32)
    for mode in ("right", "left"):
        ms, me, mp = N.merge_segments(s, e, ph, mode)
        assert [(round(a, 6), round(b, 6), names[p]) for a, b, p in zip(ms, me, mp)] == [(0, 0.2, "a"), (0.2, 0.5, "b"), (0.5, 0.6, "a")]
    ms, me, mp = N.merge_segments(s, e, ph, "previous")
    assert [(round(a, 6), round(b, 6), names[p]) for a, b, p in zip(ms, me, mp)] == [(0, 0.5, "a"), (0.5, 0.6, "a")]
    text = N.format_lab([0.009775376, 0.29], [0.1915219, 0.3], [0, 1], ["d", "SP"])
    assert text == b"97753 1915219 d\n2900000 3000000 SP\n"


def test_golden_fixture_cases():
    """every decode / merge / median / .lab case of the reference-generated fixture, through the native path"""
    with open(GOLD) as f:
        gold = json.load(f)
    labels = gold["labels"]
    table = N.LabelTable(labels)
    assert gold["cases"]
    for c in gold["cases"]:
        ids = np.array([labels.index(t) for t in c["tags"]], np.int32)
        offs = np.array(c["offsets"], np.float32)
        s, e, ph = N.decode_bio_ids(ids, table)
        assert N.to_tuples(s, e, ph, table.names) == [tuple(x) for x in c["segments_no_offsets"]]
        s, e, ph = N.decode_bio_ids(ids, table, 0.02, offs)
        assert N.to_tuples(s, e, ph, table.names) == [tuple(x) for x in c["segments"]]
        assert N.format_lab(s, e, ph, table.names).decode("utf-8") == c["lab"]
        for mode, want in c["merged"].items():
            ms, me, mp = N.merge_segments(s, e, ph, mode)
            assert N.to_tuples(ms, me, mp, table.names) == [tuple(x) for x in want], mode
        for k, want in c["median"].items():
            assert N.median_filter_ids(c["ids"], int(k)).tolist() == want


# ---- native audio ingest (wfl_host_load_wav[s]) against audio.py's read_wav + peak_normalize

def _write(path, fmt_tag, ch, sr, bits, payload, extensible=False):
    import struct
    if extensible:
        fmt = struct.pack("<HHIIHH", 0xFFFE, ch, sr, sr * ch * bits // 8, ch * bits // 8, bits) + struct.pack("<HHI", 22, bits, 0) + \
            struct.pack("<H", fmt_tag) + bytes(14)
    else:
        fmt = struct.pack("<HHIIHH", fmt_tag, ch, sr, sr * ch * bits // 8, ch * bits // 8, bits)
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"LIST" + struct.pack("<I", 3) + b"abc" + bytes(1) + \
        b"data" + struct.pack("<I", len(payload)) + payload
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", len(body)) + body)


@pytest.mark.parametrize("kind", ["pcm16", "pcm16_stereo", "pcm24", "pcm32", "pcm8", "f32", "f64_ext", "pcm16_44k"])
def test_native_wav_ingest_matches_python(kind, tmp_path):
    from wfl_asr_amd import audio as A
    rng = np.random.default_rng(len(kind))
    n = 5000
    x = rng.uniform(-0.7, 0.7, n)
    path = str(tmp_path / f"{kind}.wav")
    sr = 44100 if kind.endswith("44k") else 16000
    if kind in ("pcm16", "pcm16_44k"):
        _write(path, 1, 1, sr, 16, np.round(x * 32767).astype("<i2").tobytes())
    elif kind == "pcm16_stereo":
        y = rng.uniform(-0.5, 0.5, n)
        _write(path, 1, 2, sr, 16, np.stack([np.round(x * 32767), np.round(y * 32767)], 1).astype("<i2").tobytes())
    elif kind == "pcm24":
        v = np.round(x * 8388607).astype(np.int32)
        b = np.stack([v & 255, (v >> 8) & 255, (v >> 16) & 255], 1).astype(np.uint8)
        _write(path, 1, 1, sr, 24, b.tobytes())
    elif kind == "pcm32":
        _write(path, 1, 1, sr, 32, np.round(x * 2147483647).astype("<i4").tobytes())
    elif kind == "pcm8":
        _write(path, 1, 1, sr, 8, np.round(x * 127 + 128).astype(np.uint8).tobytes())
    elif kind == "f32":
        _write(path, 3, 1, sr, 32, x.astype("<f4").tobytes())
    elif kind == "f64_ext":
        _write(path, 3, 1, sr, 64, x.astype("<f8").tobytes(), extensible=True)
    want, want_sr = A.read_wav(path)
    want = np.asarray(A.peak_normalize(want), np.float32)
    rows = np.full((2, n + 7), 9.0, np.float32)
    ns, srs, st = A.load_wavs_into([path, path], rows, n + 7, threads=2)
    assert list(st) == [0, 0] and list(srs) == [want_sr, want_sr] and list(ns) == [want.size] * 2
    assert np.array_equal(rows[0, :want.size], want) and np.array_equal(rows[1, :want.size], want)
    assert (rows[:, want.size:] == 9.0).all()                    # nothing written past the clip


def test_native_wav_ingest_status_codes(tmp_path):
    from wfl_asr_amd import audio as A
    x = np.zeros(100, "<i2")
    p3 = str(tmp_path / "three.wav"); _write(p3, 1, 3, 16000, 16, np.zeros(300, "<i2").tobytes())
    plong = str(tmp_path / "long.wav"); _write(plong, 1, 1, 16000, 16, np.zeros(500, "<i2").tobytes())
    pbad = str(tmp_path / "bad.wav"); open(pbad, "wb").write(b"not a wav file at all")
    pmu = str(tmp_path / "mulaw.wav"); _write(pmu, 7, 1, 16000, 8, bytes(100))
    rows = np.zeros((5, 128), np.float32)
    ns, srs, st = A.load_wavs_into([p3, plong, pbad, pmu, str(tmp_path / "missing.wav")], rows, 128, threads=3)
    assert list(st) == [2, 3, 1, 1, 4] and ns[1] == 500
    assert (rows == 0).all()

"""CPU: audio ingest (WAV decode, resampling invariants, normalisation, chunking).  torchaudio/soundfile are absent,
so resample parity with torchaudio is unpinned; these tests hold the restated algorithm to its own properties."""
import struct

import numpy as np
import pytest

from wfl_asr_amd import audio as A


def test_wav_roundtrip_pcm16(tmp_path):
    x = np.sin(np.arange(4000) * 0.05) * 0.7
    p = str(tmp_path / "a.wav")
    A.write_wav(p, x, 16000)
    y, sr = A.read_wav(p)
    assert sr == 16000 and y.dtype == np.float64 and len(y) == len(x)
    assert np.abs(y - x).max() <= 1.0 / 32768 + 1e-12


def _raw_wav(path, fmt_tag, ch, sr, bits, payload, extensible=False):
    if extensible:
        fmt = struct.pack("<HHIIHH", 0xFFFE, ch, sr, sr * ch * bits // 8, ch * bits // 8, bits) + struct.pack("<HHI", 22, bits, 0) \
            + struct.pack("<H", fmt_tag) + b"\x00" * 14
    else:
        fmt = struct.pack("<HHIIHH", fmt_tag, ch, sr, sr * ch * bits // 8, ch * bits // 8, bits)
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"LIST" + struct.pack("<I", 4) + b"abcd" \
        + b"data" + struct.pack("<I", len(payload)) + payload
    open(path, "wb").write(b"RIFF" + struct.pack("<I", len(body)) + body)


def test_wav_formats(tmp_path):
    x = (np.arange(-50, 50) / 64.0).astype(np.float32)
    p = str(tmp_path / "f.wav")
    _raw_wav(p, 3, 1, 22050, 32, x.tobytes())
    y, sr = A.read_wav(p)
    assert sr == 22050 and np.array_equal(y, x.astype(np.float64))
    _raw_wav(p, 3, 1, 8000, 32, x.tobytes(), extensible=True)
    assert np.array_equal(A.read_wav(p)[0], x.astype(np.float64))
    i24 = np.array([0, 1, -1, 8388607, -8388608], dtype=np.int32)
    raw = b"".join(int(v & 0xFFFFFF).to_bytes(3, "little") for v in i24)
    _raw_wav(p, 1, 1, 16000, 24, raw)
    np.testing.assert_allclose(A.read_wav(p)[0], i24 / 8388608.0)
    st = np.stack([np.full(10, 1000, "<i2"), np.full(10, -3000, "<i2")], axis=1)
    _raw_wav(p, 1, 2, 16000, 16, st.tobytes())
    np.testing.assert_allclose(A.read_wav(p)[0], np.full(10, -1000 / 32768.0))
    with pytest.raises(ValueError):
        open(p, "wb").write(b"not a wav file")
        A.read_wav(p)


@pytest.mark.parametrize("orig,new", [(44100, 16000), (48000, 16000), (8000, 16000), (22050, 16000)])
def test_resample_properties(orig, new):
    n = orig  # one second
    t = np.arange(n) / orig
    x = 0.5 * np.sin(2 * np.pi * 440.0 * t)
    y = A.resample(x, orig, new)
    assert len(y) == int(np.ceil(new * n / orig))
    ty = np.arange(len(y)) / new
    mid = slice(len(y) // 10, -len(y) // 10)
    assert np.abs(y[mid] - 0.5 * np.sin(2 * np.pi * 440.0 * ty[mid])).max() < 2e-3     # same tone, same phase
    dc = A.resample(np.ones(n), orig, new)
    assert np.abs(dc[mid] - 1.0).max() < 2e-3                                            # unit DC gain
    if new < orig:                                                                        # content above new Nyquist is removed
        hi = A.resample(np.sin(2 * np.pi * (new * 0.75) * t), orig, new)
        assert np.abs(hi[mid]).max() < 0.02
    assert np.array_equal(A.resample(x, new, new), x)


def test_normalise_and_chunking():
    x = np.array([0.1, -0.5, 0.25])
    np.testing.assert_allclose(A.peak_normalize(x), x / (0.5 + 1e-8), rtol=0, atol=0)
    assert A.peak_normalize(np.zeros(0)).size == 0
    sr = 16000
    short = A.chunk_clip(np.ones(30 * sr) * 0.5, sr)            # exactly 30 s is NOT split (infer.py:237 uses >)
    assert len(short) == 1 and short[0].dtype == np.float32 and len(short[0]) == 30 * sr
    long = np.concatenate([np.full(30 * sr, 0.2), np.full(30 * sr, 0.4), np.full(5 * sr, 0.1)])
    ch = A.chunk_clip(A.peak_normalize(long), sr)
    assert [len(c) for c in ch] == [480000, 480000, 80000]
    for c in ch:                                                 # every chunk re-normalised to peak 1 (infer.py:115)
        assert abs(float(np.abs(c).max()) - 1.0) < 1e-6


def test_native_items_equal_python_path(tmp_path):
    """wfl_host_load_wav_chunks (decode, resample, whole-clip normalise, 30 s chunks, per-chunk re-normalise) against
    chunk_clip(load_clip(path)): bit-identical at 16 kHz, within float64 summation order after resampling."""
    rng = np.random.RandomState(5)
    sr = 16000
    cases = {
        "short.wav": (rng.randn(sr * 3) * 0.1, 16000),
        "exact30.wav": (rng.randn(sr * 30) * 0.05, 16000),
        "long.wav": (rng.randn(sr * 65) * 0.07, 16000),
        "hi.wav": (rng.randn(44100 * 2) * 0.1, 44100),
        "lo.wav": (rng.randn(8000 * 2) * 0.1, 8000),
        "longhi.wav": (rng.randn(22050 * 40) * 0.1, 22050),
    }
    for name, (x, rate) in cases.items():
        p = str(tmp_path / name)
        A.write_wav(p, x, rate)
        ref = A.chunk_clip(A.load_clip(p, sr), sr)
        got = A.load_items(p, sr)
        assert got is not None and [len(g) for g in got] == [len(r) for r in ref], name
        for g, r in zip(got, ref):
            assert g.dtype == np.float32
            if rate == sr:
                assert np.array_equal(g, r), name
            else:
                assert np.abs(g - r).max() <= 2e-7, name             # a float32 ulp at |x| <= 1
    # stereo float file, and an encoding the native decoder refuses
    st = np.stack([np.full(100, 0.25, np.float32), np.full(100, -0.75, np.float32)], axis=1)
    p = str(tmp_path / "st.wav")
    _raw_wav(p, 3, 2, 16000, 32, st.tobytes())
    assert np.array_equal(A.load_items(p)[0], A.chunk_clip(A.load_clip(p))[0])
    _raw_wav(p, 1, 3, 16000, 16, np.zeros(30, "<i2").tobytes())
    assert A.load_items(p) is None
    assert A.load_items(str(tmp_path / "missing.wav")) is None


def test_native_decoder_leaves_non_finite_samples_to_the_python_path(tmp_path):
    """A float WAV holding +-Inf or NaN: the native resampler's zero-padded taps would spread Inf * 0.0 = NaN into outputs the sample
    must not reach, so the native decoder refuses the file (status 1 -> None) and the caller falls back to audio.py, which follows
    numpy exactly; a merely large finite sample is still taken and equals the Python path."""
    rng = np.random.RandomState(9)
    x = (rng.randn(44100) * 0.1).astype(np.float32)
    for bad in (np.inf, -np.inf, np.nan):
        y = x.copy()
        y[20000] = bad
        p = str(tmp_path / "bad.wav")
        _raw_wav(p, 3, 1, 44100, 32, y.tobytes())
        assert A.load_items(p, 16000) is None
        _raw_wav(p, 3, 1, 16000, 32, y.tobytes())
        assert A.load_items(p, 16000) is None
    y = x.copy()
    y[20000] = 3.0e30
    p = str(tmp_path / "big.wav")
    _raw_wav(p, 3, 1, 44100, 32, y.tobytes())
    got, ref = A.load_items(p, 16000), A.chunk_clip(A.load_clip(p, 16000), 16000)
    assert got is not None and len(got) == len(ref) == 1 and np.abs(got[0] - ref[0]).max() <= 2e-7


def test_native_lab_text_equals_save_lab(tmp_path):
    from wfl_asr_amd import native_post as npost
    from wfl_asr_amd import postprocess as pp
    segs = [(0.009775376, 0.1915219, "d"), (0.29, 0.3, "SP"), (1.23456789, 2.0, "あ"), (2.0, 2.0000001, "d")]
    p = str(tmp_path / "x.lab")
    pp.save_lab(p, segs)
    assert npost.format_lab_tuples(segs) == open(p, "rb").read()
    assert npost.format_lab_tuples([]) == b""


def test_wav_sample_rate_reads_the_header_only(tmp_path):
    p = str(tmp_path / "a.wav")
    A.write_wav(p, np.zeros(100, np.float32), 44100)
    assert A.wav_sample_rate(p) == 44100
    A.write_wav(p, np.zeros(100, np.float32), 16000)
    assert A.wav_sample_rate(p) == 16000
    # a chunk in front of `fmt ` is skipped; no fmt chunk in the first 4 KiB, a truncated or foreign file, a missing file: None
    raw = open(p, "rb").read()
    with open(p, "wb") as f:
        f.write(raw[:12] + b"LIST" + (6).to_bytes(4, "little") + b"abcdef" + raw[12:])
    assert A.wav_sample_rate(p) == 16000
    with open(p, "wb") as f:
        f.write(raw[:20])
    assert A.wav_sample_rate(p) is None
    with open(p, "wb") as f:
        f.write(b"not a wav file at all")
    assert A.wav_sample_rate(p) is None
    assert A.wav_sample_rate(str(tmp_path / "missing.wav")) is None


def test_pcm16_reader_and_header_for_the_gpu_ingest_path(tmp_path):
    """wfl_host_read_pcm16: the file's 16-bit samples as they are (interleaved), chunks before `data` skipped, odd chunk sizes padded;
    other encodings / more than two channels / rows that are too short are turned away with the documented status."""
    rng = np.random.RandomState(3)
    x = (rng.randn(5000, 2) * 8000).astype("<i2")
    p = str(tmp_path / "st.wav")
    pcm = x.tobytes()
    body = b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, 1, 2, 44100, 44100 * 4, 4, 16) + b"LIST" + struct.pack("<I", 5) + b"abcde\x00" \
        + b"data" + struct.pack("<I", len(pcm)) + pcm
    open(p, "wb").write(b"RIFF" + struct.pack("<I", len(body)) + body)
    assert A.wav_header(p) == (1, 2, 44100, 16, len(pcm))
    pf = str(tmp_path / "f32.wav")
    _raw_wav(pf, 3, 1, 22050, 32, np.zeros(10, np.float32).tobytes())
    p3 = str(tmp_path / "c3.wav")
    _raw_wav(p3, 1, 3, 16000, 16, np.zeros(30, "<i2").tobytes())
    rows = np.full((5, 12000), 77, np.int16)
    nf, ch, sr, st = A.read_pcm16_into([p, pf, p3, str(tmp_path / "missing.wav"), p], rows, 12000, threads=3)
    assert list(st) == [0, 1, 2, 4, 0] and list(nf[[0, 4]]) == [5000, 5000] and list(ch[[0, 4]]) == [2, 2] and sr[0] == 44100
    assert np.array_equal(rows[0, :10000].reshape(5000, 2), x) and np.array_equal(rows[4, :10000], rows[0, :10000])
    assert (rows[0, 10000:] == 77).all()
    nf, ch, sr, st = A.read_pcm16_into([p], rows, 9999, threads=1)             # one sample short
    assert list(st) == [3] and nf[0] == 5000
    assert A.wav_header(str(tmp_path / "missing.wav")) is None

"""Audio ingest for the labeling loop: WAV decode, resample to 16 kHz, peak normalisation, 30 s chunking.

Replaces /root/reference/infer.py:217-220 (soundfile.read + torchaudio.functional.resample), :234-235 / :115
(peak normalisation, float64) and :19-28 (split_audio).  soundfile and torchaudio are not available in this
image, so decode and resampling are restated here:

  * read_wav: RIFF/WAVE PCM 8/16/24/32-bit and IEEE float 32/64 (plain and WAVE_FORMAT_EXTENSIBLE), returning
    float64 in [-1, 1) scaled like libsndfile (int16 / 32768 ...).  Multi-channel files are averaged to mono (the
    reference would fail on them: `max(abs(audio))` on a 2-D array).
  * resample: torchaudio.functional.resample's algorithm (sinc interpolation, Hann window, lowpass_filter_width 6,
    rolloff 0.99) written out from its published definition.  Parity with torchaudio is UNPINNED (the library is
    absent here and the reference holds no resampled fixtures); only its own invariants are tested.
"""
from __future__ import annotations

import math
import os
import struct

import numpy as np

from .postprocess import MAX_SEGMENT_DURATION, split_audio  # noqa: F401  (re-exported)


def wav_sample_rate(path: str):
    """Sample rate from the header alone (the first 4 KiB), or None when no `fmt ` chunk is found there / the file is not RIFF/WAVE."""
    try:
        with open(path, "rb") as f:
            head = f.read(4096)
    except OSError:
        return None
    if len(head) < 12 or head[:4] != b"RIFF" or head[8:12] != b"WAVE":
        return None
    pos = 12
    while pos + 8 <= len(head):
        cid = head[pos:pos + 4]
        size = struct.unpack("<I", head[pos + 4:pos + 8])[0]
        if cid == b"fmt ":
            if pos + 8 + 16 > len(head):
                return None
            return int(struct.unpack("<HHIIHH", head[pos + 8:pos + 24])[2])
        pos += 8 + size + (size & 1)
    return None


def wav_header(path: str):
    """(format tag, channels, sample rate, bits per sample, data bytes) from the header alone (the first 4 KiB), or None when the `fmt `
    and `data` chunk headers are not both found there / the file is not RIFF/WAVE.  (WAVE_FORMAT_EXTENSIBLE: the sub-format's tag.)"""
    try:
        with open(path, "rb") as f:
            head = f.read(4096)
    except OSError:
        return None
    if len(head) < 12 or head[:4] != b"RIFF" or head[8:12] != b"WAVE":
        return None
    pos, fmt = 12, None
    while pos + 8 <= len(head):
        cid = head[pos:pos + 4]
        size = struct.unpack("<I", head[pos + 4:pos + 8])[0]
        if cid == b"fmt ":
            if pos + 8 + 16 > len(head):
                return None
            tag, ch, sr, _, _, bits = struct.unpack("<HHIIHH", head[pos + 8:pos + 24])
            if tag == 0xFFFE and size >= 26 and pos + 8 + 26 <= len(head):
                tag = struct.unpack("<H", head[pos + 32:pos + 34])[0]
            fmt = (int(tag), int(ch), int(sr), int(bits))
        elif cid == b"data":
            return None if fmt is None else fmt + (int(size),)
        pos += 8 + size + (size & 1)
    return None


def read_pcm16_into(paths, rows, cap: int, threads: int = 8):
    """The 16-bit PCM samples of a batch of WAV files as they are (interleaved channels) into rows[i, :n_frames * channels] of an int16
    [B, ld] buffer (normally pinned memory) -- the host side of the GPU ingest path (csrc/resample.hip).  -> (n_frames, channels,
    sample_rates, status) int32 arrays; status 0 = done (include/wfl_asr.h: wfl_host_read_pcm16)."""
    import ctypes as C

    from . import _lib
    lib = _lib.load()
    n = len(paths)
    enc = [os.fsencode(p) for p in paths]
    arr = (C.c_char_p * max(n, 1))(*enc)
    nf = np.zeros(max(n, 1), np.int32)
    ch = np.zeros(max(n, 1), np.int32)
    srs = np.zeros(max(n, 1), np.int32)
    st = np.zeros(max(n, 1), np.int32)
    if hasattr(rows, "data_ptr"):
        ptr, ld = rows.data_ptr(), rows.stride(0)
    else:
        ptr, ld = rows.ctypes.data, rows.strides[0] // 2
    rc = lib.wfl_host_read_pcm16(arr, n, C.c_void_p(ptr), ld, int(cap), nf.ctypes.data_as(C.c_void_p), ch.ctypes.data_as(C.c_void_p),
                                 srs.ctypes.data_as(C.c_void_p), st.ctypes.data_as(C.c_void_p), int(threads))
    _lib.check(rc, "wfl_host_read_pcm16")
    return nf[:n], ch[:n], srs[:n], st[:n]


def read_wav(path: str):
    """-> (float64 mono samples, sample_rate)."""
    with open(path, "rb") as f:
        data = f.read()
    if len(data) < 12 or data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos = 12
    fmt = None
    pcm = None
    while pos + 8 <= len(data):
        cid = data[pos:pos + 4]
        size = struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            tag, ch, sr, _, _, bits = struct.unpack("<HHIIHH", body[:16])
            if tag == 0xFFFE and len(body) >= 26:        # WAVE_FORMAT_EXTENSIBLE: sub-format GUID's first 2 bytes
                tag = struct.unpack("<H", body[24:26])[0]
            fmt = (tag, ch, sr, bits)
        elif cid == b"data":
            pcm = body
        pos += 8 + size + (size & 1)
    if fmt is None or pcm is None:
        raise ValueError(f"{path}: missing fmt/data chunk")
    tag, ch, sr, bits = fmt
    if tag == 1:
        if bits == 8:
            x = (np.frombuffer(pcm, dtype=np.uint8).astype(np.float64) - 128.0) / 128.0
        elif bits == 16:
            x = np.frombuffer(pcm[:len(pcm) // 2 * 2], dtype="<i2").astype(np.float64) / 32768.0
        elif bits == 24:
            b = np.frombuffer(pcm[:len(pcm) // 3 * 3], dtype=np.uint8).reshape(-1, 3).astype(np.int32)
            v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
            v = np.where(v & 0x800000, v - 0x1000000, v)
            x = v.astype(np.float64) / 8388608.0
        elif bits == 32:
            x = np.frombuffer(pcm[:len(pcm) // 4 * 4], dtype="<i4").astype(np.float64) / 2147483648.0
        else:
            raise ValueError(f"{path}: unsupported PCM width {bits}")
    elif tag == 3:
        dt = {32: "<f4", 64: "<f8"}.get(bits)
        if dt is None:
            raise ValueError(f"{path}: unsupported float width {bits}")
        n = len(pcm) // (bits // 8) * (bits // 8)
        x = np.frombuffer(pcm[:n], dtype=dt).astype(np.float64)
    else:
        raise ValueError(f"{path}: unsupported WAVE format tag {tag}")
    if ch > 1:
        x = x[:len(x) // ch * ch].reshape(-1, ch).mean(axis=1)
    return x, int(sr)


def write_wav(path: str, samples, sr: int = 16000):
    """16-bit PCM mono writer (tests and demos)."""
    x = np.clip(np.asarray(samples, dtype=np.float64), -1.0, 32767.0 / 32768.0)
    pcm = np.round(x * 32768.0).astype("<i2").tobytes()
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(pcm)) + b"WAVE")
        f.write(b"fmt " + struct.pack("<IHHIIHH", 16, 1, 1, sr, sr * 2, 2, 16))
        f.write(b"data" + struct.pack("<I", len(pcm)) + pcm)


def resample(x: np.ndarray, orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99) -> np.ndarray:
    """Band-limited sinc resampling (Hann-windowed), float64 in / float64 out."""
    x = np.asarray(x, dtype=np.float64)
    if orig_freq == new_freq or x.size == 0:
        return x.copy()
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base = min(orig, new) * rolloff
    width = int(math.ceil(lowpass_filter_width * orig / base))
    idx = np.arange(-width, width + orig, dtype=np.float64)[None, :] / orig
    t = (np.arange(0, -new, -1, dtype=np.float64)[:, None] / new + idx) * base
    t = np.clip(t, -lowpass_filter_width, lowpass_filter_width)
    window = np.cos(t * math.pi / lowpass_filter_width / 2.0) ** 2
    t = t * math.pi
    scale = base / orig
    kernels = np.where(t == 0.0, 1.0, np.sin(t) / np.where(t == 0.0, 1.0, t)) * window * scale   # [new, 2*width+orig]
    length = x.shape[0]
    xp = np.pad(x, (width, width + orig))
    klen = kernels.shape[1]
    nfr = (xp.shape[0] - klen) // orig + 1
    frames = np.lib.stride_tricks.sliding_window_view(xp, klen)[::orig][:nfr]                 # [frames, klen]
    out = (frames @ kernels.T).reshape(-1)                                                      # frame-major, phase-minor
    target = int(math.ceil(new * length / orig))
    return out[:target]


def peak_normalize(x: np.ndarray) -> np.ndarray:
    """audio / (max|audio| + 1e-8) in float64 (infer.py:234-235; the reference's Python `max` loop, vectorised)."""
    x = np.asarray(x, dtype=np.float64)
    if x.size == 0:
        return x
    return x / (np.max(np.abs(x)) + 1e-8)


def load_clip(path: str, sample_rate: int = 16000):
    """decode -> resample to `sample_rate` -> peak-normalise; returns float64 samples (as infer.py holds them)."""
    x, sr = read_wav(path)
    if sr != sample_rate:
        x = resample(x, sr, sample_rate)
    return peak_normalize(x)


def load_wavs_into(paths, rows, cap: int, threads: int = 8):
    """Native, threaded fast path of `load_clip` + float32 cast for a batch of files (csrc/hostpost.hip through the C ABI):
    file i is decoded, mixed to mono, peak-normalised in float64 and stored as float32 into rows[i, :n] (a torch / numpy
    float32 [B, ld] buffer, normally pinned memory).  -> (n_samples, sample_rates, status) int32 arrays; status 0 = done,
    anything else = the caller must take the Python path for that file (see include/wfl_asr.h).  Bit-identical to
    `np.asarray(load_clip(path), np.float32)` for the files it accepts."""
    import ctypes as C

    from . import _lib
    lib = _lib.load()
    n = len(paths)
    enc = [os.fsencode(p) for p in paths]
    arr = (C.c_char_p * max(n, 1))(*enc)
    ns = np.zeros(max(n, 1), np.int32)
    srs = np.zeros(max(n, 1), np.int32)
    st = np.zeros(max(n, 1), np.int32)
    if hasattr(rows, "data_ptr"):
        ptr, ld = rows.data_ptr(), rows.stride(0)
    else:
        ptr, ld = rows.ctypes.data, rows.strides[0] // 4
    rc = lib.wfl_host_load_wavs(arr, n, C.c_void_p(ptr), ld, int(cap), ns.ctypes.data_as(C.c_void_p),
                                srs.ctypes.data_as(C.c_void_p), st.ctypes.data_as(C.c_void_p), int(threads))
    _lib.check(rc, "wfl_host_load_wavs")
    return ns[:n], srs[:n], st[:n]


def load_items(path: str, sample_rate: int = 16000):
    """The general ingest path of one file in native code (csrc/hostpost.hip: wfl_host_load_wav_chunks): decode, resample,
    whole-clip peak normalisation, 30 s chunking with per-chunk re-normalisation -> list of float32 work items, exactly what
    `chunk_clip(load_clip(path))` returns (bit-identical without resampling; resampled clips agree to ~1e-12 before the
    float32 cast -- numpy's BLAS sums in another order).  Returns None when the native decoder does not take the file
    (unsupported encoding, > 2 channels): the caller falls back to the Python path."""
    import ctypes as C

    from . import _lib
    lib = _lib.load()
    chunk = int(MAX_SEGMENT_DURATION * sample_rate)
    rows = 4                       # (pages of rows that stay unused are never touched; the library reports how many it needs beyond)
    while True:
        buf = np.empty((rows, chunk), np.float32)
        n_rows, sr = C.c_int32(0), C.c_int32(0)
        lens = np.zeros(rows, np.int32)
        st = lib.wfl_host_load_wav_chunks(os.fsencode(path), int(sample_rate), chunk, buf.ctypes.data_as(C.c_void_p), chunk, rows,
                                          C.byref(n_rows), lens.ctypes.data_as(C.c_void_p), C.byref(sr))
        if st == 5:
            rows = int(n_rows.value)
            continue
        if st != 0:
            return None
        return [buf[r, :lens[r]] for r in range(n_rows.value)]       # views: `buf` belongs to this call alone


def chunk_clip(audio: np.ndarray, sr: int = 16000):
    """The reference's two cases (infer.py:237-244): <= 30 s -> one item as is; longer -> non-overlapping 30 s chunks,
    each re-normalised (process_segments, infer.py:114-115).  Returns float32 arrays (infer.py:134, 251)."""
    if len(audio) / sr > MAX_SEGMENT_DURATION:
        return [(peak_normalize(seg) if len(seg) > 0 else seg).astype(np.float32) for seg in split_audio(audio, sr)]
    return [np.asarray(audio, dtype=np.float32)]

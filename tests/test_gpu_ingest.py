"""-m gpu: audio ingest on the GPU (csrc/resample.hip; SURVEY.md section 8f rank 1 -- infer.py:217-220, 234-235): 16-bit PCM rows ->
float64 sinc resampling -> peak normalisation -> float32 rows.  Held to the host loader (wfl_host_load_wav_chunks, itself held to the
Python restatement audio.py in tests/test_audio_cpu.py) bit for bit; parity with torchaudio itself is UNPINNED (library absent here,
the reference holds no resampled fixtures)."""
import ctypes as C
import os
import struct

import numpy as np
import pytest
import torch
import yaml

from wfl_asr_amd import _lib
from wfl_asr_amd import audio as A
from wfl_asr_amd import infer as I
import synthetic as synth
from cases import tiny_whisper_config

pytestmark = pytest.mark.gpu


def _write_pcm16(path, x, sr):
    """x: [n] or [n, 2] float in [-1, 1) -> 16-bit PCM WAV."""
    x = np.asarray(x)
    ch = 1 if x.ndim == 1 else x.shape[1]
    pcm = np.round(x * 32767.0).astype("<i2").tobytes()
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(pcm)) + b"WAVE")
        f.write(b"fmt " + struct.pack("<IHHIIHH", 16, 1, ch, sr, sr * 2 * ch, 2 * ch, 16))
        f.write(b"data" + struct.pack("<I", len(pcm)) + pcm)


@pytest.mark.parametrize("rate", [44100, 48000, 8000, 22050, 11025, 16000])     # (16000: decode + normalise only, round 4)
def test_gpu_resample_equals_the_host_loader_bit_for_bit(tmp_path, rate):
    lib = _lib.load()
    rng = np.random.RandomState(rate)
    secs = [2.0, 0.37, 5.5, 29.99]
    paths = []
    for i, s in enumerate(secs):
        n = int(rate * s)
        t = np.arange(n) / rate
        x = 0.4 * np.sin(2 * np.pi * (200.0 + 150 * i) * t) + 0.1 * rng.randn(n)
        x = np.clip(x, -0.99, 0.99)
        if i % 2:
            x = np.stack([x, 0.5 * np.roll(x, 7)], axis=1)                 # two channels
        p = str(tmp_path / f"c{i}.wav")
        _write_pcm16(p, x, rate)
        paths.append(p)
    B, L = len(paths), 480000
    cap_in = 2 * (int(np.ceil(L * rate / 16000)) + 2)
    rows = torch.zeros(B, cap_in, dtype=torch.int16).pin_memory()
    nf, ch, srs, st = A.read_pcm16_into(paths, rows, cap_in, threads=4)
    assert list(st) == [0] * B and list(srs) == [rate] * B and list(ch) == [1, 2, 1, 2]
    d_rows = rows.cuda()
    d_nf, d_ch = torch.from_numpy(nf.copy()).cuda(), torch.from_numpy(ch.copy()).cuda()
    out = torch.full((B, L), 7.0, dtype=torch.float32, device="cuda")
    wsb = int(lib.wfl_resample_workspace_bytes(B, L))
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    rc = lib.wfl_resample_pcm16(C.c_void_p(d_rows.data_ptr()), cap_in, C.c_void_p(d_nf.data_ptr()), C.c_void_p(d_ch.data_ptr()), B, rate, 16000,
                                C.c_void_p(out.data_ptr()), L, L, C.c_void_p(ws.data_ptr()), wsb, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    for i, p in enumerate(paths):
        ref = A.load_items(p, 16000)                                        # the host loader: decode, resample, normalise
        assert ref is not None and len(ref) == 1
        n = len(ref[0])
        assert n == int(np.ceil(16000 * int(nf[i]) / rate))
        assert np.array_equal(got[i, :n], ref[0]), (rate, i, float(np.abs(got[i, :n] - ref[0]).max()))
        assert not got[i, n:].any()                                         # the row's tail is zeros
        py = A.chunk_clip(A.load_clip(p, 16000), 16000)[0]                  # the Python restatement (BLAS sums in another order)
        assert np.abs(got[i, :n] - py).max() <= 2e-7


def test_folder_at_44k_labels_like_the_host_ingest_path(tmp_path, monkeypatch):
    """Labeler.label_files over 44.1 kHz files (one of them 16 kHz, one longer than 30 s: those keep their own paths): the
    GPU ingest path gives exactly the segments of the host ingest path (WFL_GPU_INGEST=0)."""
    d = tmp_path
    cfg = tiny_whisper_config(enable_bilstm=False)
    cfg["model"]["encoder_arch"]["max_positions"] = 1500
    cfg["output"]["save_dir"] = str(d / "save")
    cfg["postprocess"] = {"median_filter": 3, "merge_segments": "right", "confidence_threshold": 0.3}
    os.makedirs(cfg["output"]["save_dir"])
    labels = synth.make_labels(5)
    with open(d / "save" / "phonemes.txt", "w") as f:
        f.write("\n".join(labels) + "\n")
    with open(d / "save" / "langs.txt", "w") as f:
        f.write("en,0\nja,1\n")
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, len(labels), seed=33).items()}
    os.makedirs(d / "wavs")
    paths = []
    for i in range(7):
        x = A.resample(synth.make_clip(800 + i, 16000 * (3 + 2 * i), seed=33).astype(np.float64), 16000, 44100) * 0.9
        p = str(d / "wavs" / f"h{i}.wav")
        _write_pcm16(p, np.clip(x, -0.99, 0.99) if i != 3 else np.stack([np.clip(x, -0.99, 0.99)] * 2, axis=1), 44100)
        paths.append(p)
    p16 = str(d / "wavs" / "k16.wav")
    A.write_wav(p16, synth.make_clip(850, 16000 * 4, seed=33) * 0.8, 16000)
    plong = str(d / "wavs" / "long44.wav")
    _write_pcm16(plong, np.clip(A.resample(synth.make_clip(851, 16000 * 40, seed=33).astype(np.float64), 16000, 44100) * 0.9, -0.99, 0.99), 44100)
    paths += [p16, plong]
    lab = I.Labeler(cfg, sd, "cuda", batch_size=4)
    calls = []
    orig = lab._label_resampled
    monkeypatch.setattr(lab, "_label_resampled", lambda *a, **k: (calls.append(len(a[0])), orig(*a, **k))[1])
    got = lab.label_files(paths, lang_id=0, confidence_threshold=0.3, verbose=False)
    assert calls == [7]                                          # the seven short 44.1 kHz files, nothing else
    monkeypatch.setenv("WFL_GPU_INGEST", "0")
    ref = lab.label_files(paths, lang_id=0, confidence_threshold=0.3, verbose=False)
    assert calls == [7]
    assert got == ref
    assert all(len(s) > 0 for s in got)

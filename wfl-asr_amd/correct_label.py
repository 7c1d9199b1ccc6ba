"""Acoustic boundary snapping of `.lab` files: the step the reference's notebook runs right after inference
(/root/reference/correct_label.py).

  detect_boundaries       correct_label.py:15-38   spectral flux of a 512-point STFT + mean |delta| of 13 MFCCs, summed half and
                                                   half, peak picking (height 0.1, distance 5 frames), peaks shifted one frame back
  correct_lab_boundaries  correct_label.py:40-87   every start / end of a `.lab` line snaps to the nearest unused detected boundary
                                                   within 30 ms (greedy, in file order, a boundary is used once)
  write_lab / process_file  correct_label.py:139-177 (truncating int(t * 1e7), like save_lab)

The reference computes the features with librosa 0.11 (`stft`, `feature.mfcc`, `feature.delta`); librosa is not available in this
image and the reference holds no fixtures for this step, so the feature arithmetic is restated from librosa's documented
definitions with numpy / scipy and its parity is UNPINNED (only its own properties are tested).  The snapping logic is exact
Python and is pinned by 220 outputs of the reference's own function (tests/golden/metrics.json) and a worked example.  The two STFTs, the mel / dB / DCT chain and the flux run on the GPU when one is there
(csrc/stft.hip through the C ABI: wfl_boundary_features; `device=None` picks it automatically, `device="cpu"` keeps the numpy
restatement, which the GPU path is tested against); the delta filter, the peak picking and the snapping stay on the host (a few
thousand values per file).  A file at another rate is resampled to 16 kHz first, as `librosa.load(path, sr=16000)` does
(correct_label.py:154) -- with this package's band-limited sinc resampler (audio.resample, the arithmetic of csrc/resample.hip), not
librosa's soxr: same band limit, different filter, parity UNPINNED like the features'.  A folder is processed by a pool of workers
like the reference's ProcessPoolExecutor (correct_label.py:200-201): threads here, so that all of them share the one GPU context.
Like the reference, an interrupted run leaves `<wav>_boundary.txt` behind and the next run reuses it (correct_label.py:89-104,
153-161); `--save_plot` is not offered (matplotlib plotting is out of scope).
"""
from __future__ import annotations

import os

import numpy as np

from . import audio as A

snap_threshold_sec = 0.03


def _hann_periodic(n: int) -> np.ndarray:
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def _stft_mag(y: np.ndarray, n_fft: int, hop: int) -> np.ndarray:
    """|librosa.stft(y, n_fft, hop)|: periodic Hann of length n_fft, center=True with zero padding (librosa >= 0.10),
    frames = 1 + len(y) // hop.  -> [1 + n_fft // 2, frames] float32."""
    y = np.asarray(y, dtype=np.float32)
    yp = np.pad(y, (n_fft // 2, n_fft // 2))
    n_frames = 1 + (len(yp) - n_fft) // hop
    frames = np.lib.stride_tricks.sliding_window_view(yp, n_fft)[::hop][:n_frames]
    spec = np.fft.rfft(frames * _hann_periodic(n_fft).astype(np.float32), axis=1)
    return np.abs(spec).T.astype(np.float32)


def _mel_slaney(sr: int, n_fft: int, n_mels: int) -> np.ndarray:
    """librosa.filters.mel(sr=sr, n_fft=n_fft, n_mels=n_mels): Slaney mel scale, Slaney area normalisation.  -> [n_mels, 1 + n_fft // 2]"""
    def hz2mel(f):
        f = np.asarray(f, dtype=np.float64)
        return np.where(f >= 1000.0, 15.0 + np.log(np.maximum(f, 1e-30) / 1000.0) * (27.0 / np.log(6.4)), 3.0 * f / 200.0)

    def mel2hz(m):
        m = np.asarray(m, dtype=np.float64)
        return np.where(m >= 15.0, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)), 200.0 * m / 3.0)

    fftfreqs = np.linspace(0.0, sr / 2.0, 1 + n_fft // 2)
    mel_f = mel2hz(np.linspace(hz2mel(0.0), hz2mel(sr / 2.0), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    lower = -ramps[:-2] / fdiff[:-1, None]
    upper = ramps[2:] / fdiff[1:, None]
    w = np.maximum(0.0, np.minimum(lower, upper))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w.astype(np.float32)


def _mfcc(y: np.ndarray, sr: int, n_mfcc: int, hop: int) -> np.ndarray:
    """librosa.feature.mfcc(y=y, sr=sr, n_mfcc=n_mfcc, hop_length=hop): power mel spectrogram (n_fft 2048, 128 mels) -> dB with
    top_db 80 -> DCT-II, orthonormal, first n_mfcc rows."""
    from scipy.fft import dct
    power = _stft_mag(y, 2048, hop).astype(np.float32) ** 2
    mel = _mel_slaney(sr, 2048, 128) @ power
    log_spec = 10.0 * np.log10(np.maximum(1e-10, mel))
    log_spec = np.maximum(log_spec, log_spec.max() - 80.0)
    return dct(log_spec, axis=0, type=2, norm="ortho")[:n_mfcc]


_GPU_TABLES = {}


def gpu_available() -> bool:
    try:
        import torch
        return bool(torch.cuda.is_available())
    except Exception:
        return False


def boundary_features_gpu(y, device="cuda"):
    """(flux before normalisation [F + 1 -> cut by the caller], mfcc [13, F]) of one 16 kHz clip from csrc/stft.hip: the same
    quantities as `_stft_mag` / `_mfcc` below, float32 on the MFMA."""
    import ctypes as C

    import torch

    from . import _lib
    lib = _lib.load()
    dev = torch.device(device)
    y = np.ascontiguousarray(np.asarray(y, dtype=np.float32))
    L = int(y.shape[0])
    if L <= 0:
        raise ValueError("empty clip")
    key = str(dev)
    if key not in _GPU_TABLES:
        from scipy.fft import dct
        mel = _mel_slaney(16000, 2048, 128).astype(np.float32)                         # [128, 1025]
        d = dct(np.eye(128), axis=0, type=2, norm="ortho")[:13].astype(np.float32)      # [13, 128]: row c = coefficient c's weights
        _GPU_TABLES[key] = (torch.from_numpy(np.ascontiguousarray(mel)).to(dev), torch.from_numpy(np.ascontiguousarray(d)).to(dev))
    mel_w, dctm = _GPU_TABLES[key]
    F = 1 + L // 160
    wav = torch.from_numpy(y).to(dev)
    flux = torch.empty(F, dtype=torch.float32, device=dev)
    mfcc = torch.empty(13, F, dtype=torch.float32, device=dev)
    nb = int(lib.wfl_boundary_workspace_bytes(1, L))
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    rc = lib.wfl_boundary_features(C.c_void_p(wav.data_ptr()), L, None, 1, L, C.c_void_p(mel_w.data_ptr()), C.c_void_p(dctm.data_ptr()),
                                   C.c_void_p(flux.data_ptr()), C.c_void_p(mfcc.data_ptr()), C.c_void_p(ws.data_ptr()), nb,
                                   C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    _lib.check(rc, "wfl_boundary_features")
    return flux.cpu().numpy(), mfcc.cpu().numpy()


def boundary_features_gpu_batch(clips, device="cuda"):
    """The same features for several 16 kHz clips of different lengths in ONE launch (wfl_boundary_features with `lens`): -> a list of
    (flux [F_b], mfcc [13, F_b]) with F_b = 1 + len_b // 160, each equal to the clip's own single-clip call."""
    import ctypes as C

    import torch

    from . import _lib
    lib = _lib.load()
    dev = torch.device(device)
    boundary_features_gpu(np.zeros(1600, np.float32), device)          # the device's tables
    mel_w, dctm = _GPU_TABLES[str(dev)]
    lens = np.array([len(c) for c in clips], np.int32)
    B, L = len(clips), int(lens.max())
    if B == 0 or int(lens.min()) <= 0:
        raise ValueError("empty batch or empty clip")
    host = np.zeros((B, L), np.float32)
    for b, c in enumerate(clips):
        host[b, :len(c)] = np.asarray(c, dtype=np.float32)
    F = 1 + L // 160
    wav = torch.from_numpy(host).to(dev)
    d_lens = torch.from_numpy(lens).to(dev)
    flux = torch.empty(B, F, dtype=torch.float32, device=dev)
    mfcc = torch.empty(B, 13, F, dtype=torch.float32, device=dev)
    nb = int(lib.wfl_boundary_workspace_bytes(B, L))
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    rc = lib.wfl_boundary_features(C.c_void_p(wav.data_ptr()), L, C.c_void_p(d_lens.data_ptr()), B, L, C.c_void_p(mel_w.data_ptr()),
                                   C.c_void_p(dctm.data_ptr()), C.c_void_p(flux.data_ptr()), C.c_void_p(mfcc.data_ptr()),
                                   C.c_void_p(ws.data_ptr()), nb, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    _lib.check(rc, "wfl_boundary_features")
    flux, mfcc = flux.cpu().numpy(), mfcc.cpu().numpy()
    return [(flux[b, :1 + int(n) // 160], mfcc[b, :, :1 + int(n) // 160]) for b, n in enumerate(lens)], flux


def detect_boundaries(y, sr, frame_length=512, hop_length=160, flux_threshold=0.1, delta_window=5, device=None):
    """-> (boundary times [s], flux, mfcc-delta magnitude, frame times), as correct_label.py:15-38.  device: None = the GPU when there
    is one (and the defaults n_fft 512 / hop 160 / 16 kHz are asked for), "cpu" = the numpy restatement."""
    from scipy.signal import find_peaks, savgol_filter
    use_gpu = device != "cpu" and frame_length == 512 and hop_length == 160 and sr == 16000 and (device is not None or gpu_available())
    if use_gpu:
        flux, mfcc = boundary_features_gpu(y, device or "cuda")
        flux = flux / np.max(flux)
    else:
        S = _stft_mag(y, frame_length, hop_length)
        flux = np.sqrt(np.sum(np.diff(S, axis=1) ** 2, axis=0))
        flux = np.pad(flux, (1,), mode="constant")
        flux = flux / np.max(flux)
        mfcc = _mfcc(y, sr, 13, hop_length)
    delta = savgol_filter(mfcc, 9, deriv=1, axis=-1, polyorder=1, mode="interp")     # librosa.feature.delta defaults
    delta_mag = np.mean(np.abs(delta), axis=0)
    delta_mag = delta_mag / np.max(delta_mag)
    n = min(len(flux), len(delta_mag))
    flux, delta_mag = flux[:n], delta_mag[:n]
    combined = 0.5 * flux + 0.5 * delta_mag
    peaks, _ = find_peaks(combined, height=flux_threshold, distance=delta_window)
    shifted = np.clip(peaks - 1, 0, len(combined) - 1)
    times = shifted * hop_length / float(sr)
    flux_times = np.arange(len(flux)) * hop_length / float(sr)
    return times.tolist(), flux, delta_mag, flux_times


def snap_segments(segments, predicted_boundaries, snap_threshold=snap_threshold_sec):
    """The loop of correct_label.py:52-85 on [(start_s, end_s, label)]: start, then end, of every line takes the nearest boundary
    not used before, if it lies within the threshold."""
    used = set()
    out = []
    for start_sec, end_sec, label in segments:
        for which in (0, 1):
            t0 = start_sec if which == 0 else end_sec
            closest, best = None, snap_threshold + 1
            for t in predicted_boundaries:
                if t in used:
                    continue
                dist = abs(t - t0)
                if dist < best:
                    best, closest = dist, t
            if closest is not None and best <= snap_threshold:
                used.add(closest)
                if which == 0:
                    start_sec = closest
                else:
                    end_sec = closest
        out.append((start_sec, end_sec, label))
    return out


def read_lab(lab_path):
    segs = []
    with open(lab_path, "r") as f:
        for line in f:
            parts = line.strip().split()
            if len(parts) == 3:
                segs.append((float(parts[0]) / 1e7, float(parts[1]) / 1e7, parts[2]))
    return segs


def correct_lab_boundaries(wav_path, predicted_boundaries, snap_threshold=snap_threshold_sec):
    """-> (snapped, original) segment lists for `<wav>.lab` (both empty when the file is missing), correct_label.py:40-87."""
    lab_path = wav_path.replace(".wav", ".lab")
    if not os.path.exists(lab_path):
        return [], []
    original = read_lab(lab_path)
    return snap_segments(original, predicted_boundaries, snap_threshold), original


def write_lab(wav_path, snapped_boundaries, out_path=None):
    lab_path = wav_path.replace(".wav", ".lab") if out_path is None else out_path
    with open(lab_path, "w") as f:
        for start, end, label in snapped_boundaries:
            f.write(f"{int(start * 1e7)} {int(end * 1e7)} {label}\n")


def process_file(wav_path, device=None):
    """correct_label.py:153-177 without the plotting: detect (or reuse `<wav>_boundary.txt`), snap, rewrite the `.lab`."""
    y, sr = A.read_wav(wav_path)
    if y.ndim > 1:
        y = y.mean(axis=1)                                # librosa.load(mono=True)
    if sr != 16000:
        y, sr = A.resample(np.asarray(y, dtype=np.float64), int(sr), 16000), 16000     # librosa.load(path, sr=16000)
    y = np.asarray(y, dtype=np.float32)
    txt = wav_path.replace(".wav", "_boundary.txt")
    if os.path.exists(txt):
        with open(txt) as f:
            predicted = [float(line.strip()) for line in f if line.strip()]
    else:
        predicted = detect_boundaries(y, sr, device=device)[0]
        with open(txt, "w") as f:                       # (correct_label.py:89-97, 160: what an interrupted run leaves behind)
            for t in predicted:
                f.write(f"{t:.6f}\n")
    snapped, _ = correct_lab_boundaries(wav_path, predicted)
    write_lab(wav_path, snapped)
    if os.path.exists(txt):
        os.remove(txt)
    return snapped


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="Correct .lab timing boundaries from audio features.")
    ap.add_argument("input_path", type=str, help="Path to .wav file or folder containing .wav files")
    ap.add_argument("--workers", type=int, default=0, help="files in flight in folder mode (default: min(8, host cores))")
    args = ap.parse_args(argv)
    if os.path.isdir(args.input_path):
        process_folder(args.input_path, workers=args.workers)
    else:
        process_file(args.input_path)


def process_folder(folder, workers=0, device=None):
    """Every `*.wav` of a folder, several files in flight (correct_label.py:196-207).  -> {path: snapped segments}; a file that fails
    is reported and skipped like the reference's `[ERROR]` line, the others still finish."""
    from concurrent.futures import ThreadPoolExecutor, as_completed
    wavs = [os.path.join(folder, f) for f in sorted(os.listdir(folder)) if f.endswith(".wav")]
    workers = workers or max(1, min(8, os.cpu_count() or 1))
    if device is None and gpu_available() and wavs:
        boundary_features_gpu(np.zeros(1600, np.float32))      # tables and kernels of the device exist before the workers start
    out = {}
    with ThreadPoolExecutor(max_workers=workers) as ex:
        futs = {ex.submit(process_file, w, device): w for w in wavs}
        for fu in as_completed(futs):
            try:
                out[futs[fu]] = fu.result()
            except Exception as e:                              # noqa: BLE001  (the reference prints and goes on)
                print(f"[ERROR] Failed to process {futs[fu]}: {e}")
    return out


if __name__ == "__main__":
    main()

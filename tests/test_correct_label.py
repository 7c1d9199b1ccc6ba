"""CPU: boundary snapping (/root/reference/correct_label.py).  The snapping loop is pinned by a worked example; the detector's
feature arithmetic is librosa's restated (librosa absent, no reference fixtures: parity UNPINNED) and is held to its own
properties on signals with known change points."""
import os

import numpy as np

from wfl_asr_amd import audio as A
from wfl_asr_amd import correct_label as CL


def test_snapping_worked_example():
    segs = [(0.10, 0.50, "a"), (0.50, 0.90, "b")]
    pred = [0.11, 0.52, 0.88, 0.505]
    # a.start 0.10 -> 0.11; a.end 0.50 -> 0.505 (nearest unused); b.start 0.50 -> 0.52 (0.505 is used); b.end 0.90 -> 0.88
    assert CL.snap_segments(segs, pred) == [(0.11, 0.505, "a"), (0.52, 0.88, "b")]
    # beyond 30 ms nothing moves; a boundary is consumed once, in file order
    assert CL.snap_segments([(0.50, 0.70, "x")], [0.54, 0.74]) == [(0.50, 0.70, "x")]
    assert CL.snap_segments([(0.50, 0.51, "x")], [0.505]) == [(0.505, 0.51, "x")]
    assert CL.snap_segments([], [0.1]) == []


def test_detector_finds_known_change_points():
    sr = 16000
    t = np.arange(sr) / sr
    y = np.concatenate([0.5 * np.sin(2 * np.pi * 300 * t), 0.5 * np.sin(2 * np.pi * 2500 * t), np.zeros(sr // 2),
                        0.4 * np.sin(2 * np.pi * 800 * t[:sr // 2])]).astype(np.float32)
    times, flux, delta_mag, ft = CL.detect_boundaries(y, sr)
    assert len(flux) == len(delta_mag) == len(ft) == 1 + len(y) // 160
    assert abs(float(flux.max()) - 1.0) < 1e-6 and abs(float(delta_mag.max()) - 1.0) < 1e-6
    for change in (1.0, 2.0, 2.5):
        assert min(abs(x - change) for x in times) <= 0.07, (change, times)     # the 128 ms MFCC window smears a change over +-64 ms
    assert all(b - a >= 5 * 160 / sr - 1e-9 for a, b in zip(times, times[1:]))     # peak distance of 5 frames


def test_process_file_rewrites_lab(tmp_path):
    sr = 16000
    t = np.arange(sr) / sr
    y = np.concatenate([0.5 * np.sin(2 * np.pi * 300 * t), 0.5 * np.sin(2 * np.pi * 2500 * t)])
    wav = str(tmp_path / "u.wav")
    A.write_wav(wav, y, sr)
    with open(tmp_path / "u.lab", "w") as f:
        f.write("0 9850000 a\n9850000 20000000 b\n")
    snapped = CL.process_file(wav)
    assert [s[2] for s in snapped] == ["a", "b"]
    assert abs(snapped[0][1] - 1.0) <= 0.02 and snapped[0][1] != 0.985          # the 0.985 s boundary moved to the tone change
    lines = open(tmp_path / "u.lab").read().split("\n")
    assert lines[0].split()[2] == "a" and int(lines[0].split()[1]) == int(snapped[0][1] * 1e7)
    # a pre-made boundary file wins over the detector and is removed afterwards (correct_label.py:157-177)
    with open(tmp_path / "u.lab", "w") as f:
        f.write("0 5000000 a\n")
    with open(tmp_path / "u_boundary.txt", "w") as f:
        f.write("0.510000\n")
    assert CL.process_file(wav) == [(0.0, 0.51, "a")]
    assert not os.path.exists(tmp_path / "u_boundary.txt")

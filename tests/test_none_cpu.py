"""CPU: `encoder_type: none` -- config resolution, C-ABI validation (wfl_create runs without a GPU) and the oracle's restatement
of torchaudio.transforms.MelSpectrogram held against an independent numpy computation (the restatement itself is PARITY UNPINNED:
torchaudio is not installed and the reference holds no fixture for this front-end)."""
import numpy as np
import pytest
import torch

from oracle import wfl_oracle as O
from wfl_asr_amd import _lib
import synthetic as synth
from wfl_asr_amd.archs import MelArch, resolve_encoder_arch
from wfl_asr_amd.tagger import BIOPhonemeTagger


def _cfg(frame_duration=0.02, n_mels=80, **kw):
    cfg = synth.base_config("none", **kw)
    cfg["data"]["frame_duration"] = frame_duration
    cfg["data"]["n_mels"] = n_mels
    return cfg


def test_config_resolution():
    cfg = _cfg()
    enc, arch = resolve_encoder_arch(cfg["model"], cfg["data"])
    assert enc == "none" and arch == MelArch(80, 80, 320, 400, 16000)
    cfg["model"]["encoder_type"] = "null"                                   # model.py:82 accepts both spellings
    assert resolve_encoder_arch(cfg["model"], cfg["data"])[0] == "none"
    with pytest.raises(ValueError):
        resolve_encoder_arch(cfg["model"])                                  # needs config["data"]
    cfg["data"]["sample_rate"] = 22050
    with pytest.raises(ValueError):
        resolve_encoder_arch(cfg["model"], cfg["data"])
    # the synthetic checkpoint of this model is the head alone, at width n_mels (model.py:91-93)
    sd = synth.make_state_dict(_cfg(), 11, seed=1)
    assert not any(k.startswith("encoder.") for k in sd)
    assert sd["lang_proj.weight"].shape == (80, 144) and sd["bilstm.weight_hh_l0"].shape == (160, 40)
    assert sd["classifier.weight"].shape == (11, 80)


def test_create_validates_the_mel_geometry():
    labels = synth.make_labels(3)
    m = BIOPhonemeTagger(_cfg(), labels)                                    # 80 bins, hop 320, 2 heads: fine
    assert m.hidden_size == 80 and m.encoder_type == "none"
    assert m.num_frames(480000) == 1501 and m.num_frames(16000) == 51 and m.num_frames(200) == 0
    BIOPhonemeTagger(_cfg(0.01), labels)                                    # hop 160
    with pytest.raises(_lib.WflError, match="hop 160 and 320"):
        BIOPhonemeTagger(_cfg(0.0125), labels)
    with pytest.raises(_lib.WflError, match="conformer_heads"):
        BIOPhonemeTagger(_cfg(conformer_heads=3), labels)                   # 80 % 3 != 0: nn.MultiheadAttention refuses it too
    with pytest.raises(_lib.WflError, match="odd n_mels"):
        BIOPhonemeTagger(_cfg(n_mels=81, num_conformer_layers=0), labels)


def test_oracle_mel_against_plain_numpy():
    rng = np.random.default_rng(5)
    L, hop, n_mels = 4000, 320, 80
    x = rng.standard_normal((2, L)).astype(np.float32)
    got = O.mel_spectrogram_power(torch.from_numpy(x), 16000, 400, hop, n_mels).numpy()
    assert got.shape == (2, n_mels, 1 + L // hop)
    # independent: explicit reflect padding, periodic Hann, rfft, |.|^2, HTK triangles in float64
    win = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(400) / 400)
    xp = np.pad(x.astype(np.float64), ((0, 0), (200, 200)), mode="reflect")
    frames = np.stack([xp[:, t * hop:t * hop + 400] * win for t in range(1 + L // hop)], axis=1)
    power = np.abs(np.fft.rfft(frames, axis=-1)) ** 2                        # [B, T, 201]
    hz2mel = lambda f: 2595.0 * np.log10(1.0 + f / 700.0)
    mel2hz = lambda m: 700.0 * (10.0 ** (m / 2595.0) - 1.0)
    pts = mel2hz(np.linspace(hz2mel(0.0), hz2mel(8000.0), n_mels + 2))
    freqs = np.linspace(0, 8000, 201)
    fb = np.zeros((201, n_mels))
    for i in range(n_mels):
        down = (freqs - pts[i]) / (pts[i + 1] - pts[i])
        up = (pts[i + 2] - freqs) / (pts[i + 2] - pts[i + 1])
        fb[:, i] = np.maximum(0.0, np.minimum(down, up))
    ref = (power @ fb).transpose(0, 2, 1)
    assert np.abs(got - ref).max() <= 2e-4 * ref.max()
    assert np.abs(O.mel_filter_bank_htk(n_mels) - fb).max() <= 2e-5
    # every band sees at least one bin at 80 bands / 201 bins (no dead hidden channel)
    assert (fb.sum(0) > 0).all()


def test_batches_in_flight_default_and_override(monkeypatch):
    """tagger.batches_in_flight: 3 for a BiLSTM behind Whisper-tiny / -base, 2 everywhere else; WFL_INFLIGHT overrides."""
    labels = synth.make_labels(3)
    monkeypatch.delenv("WFL_INFLIGHT", raising=False)
    assert BIOPhonemeTagger(synth.base_config("whisper"), labels).batches_in_flight() == 3            # default config.yaml head
    assert BIOPhonemeTagger(synth.baseline_config(1), labels).batches_in_flight() == 2                # cfg2: no BiLSTM
    assert BIOPhonemeTagger(synth.baseline_config(2), labels).batches_in_flight() == 2                # WavLM-large + BiLSTM
    assert BIOPhonemeTagger(synth.baseline_config(3), labels).batches_in_flight() == 2                # Whisper-small + full head
    assert BIOPhonemeTagger(_cfg(), labels).batches_in_flight() == 2                                  # mel front-end
    monkeypatch.setenv("WFL_INFLIGHT", "4")
    assert BIOPhonemeTagger(synth.baseline_config(1), labels).batches_in_flight() == 4

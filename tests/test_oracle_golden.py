"""Pin the oracle (oracle/wfl_oracle.py) to outputs of the reference itself (tests/golden/*.npz,
produced in the build container by tests/golden/make_golden.py from /root/reference + HF transformers)."""
import os

import numpy as np
import pytest
import torch

from oracle import wfl_oracle as O
import synthetic as synth
from wfl_asr_amd.archs import resolve_encoder_arch
from cases import GOLDEN_CASES


def _run(name, golden_dir):
    path = os.path.join(golden_dir, name + ".npz")
    if not os.path.exists(path):
        pytest.skip(f"{name}.npz not generated")
    g = np.load(path)
    cfg = GOLDEN_CASES[name]()
    labels = synth.make_labels(int(g["n_phonemes"]))
    enc, arch = resolve_encoder_arch(cfg["model"])
    sd_np = synth.make_state_dict(cfg, len(labels), seed=int(g["seed"]))
    if "bf16_weights" in g and int(g["bf16_weights"]):
        sd_np = synth.round_weights_bf16(sd_np)
    sd = O.to_torch_state_dict(sd_np)
    B, L = len(g["lang_id"]), int(g["L"])
    if name == "wavlm_base_cfg1":
        wav = np.stack([synth.sine_clip(L)] * B)
    else:
        wav = synth.make_batch(int(g["clip0"]), B, L, seed=int(g["seed"]))
    wav = torch.from_numpy(wav)
    logits, offs, hid = O.forward(wav, torch.from_numpy(g["lang_id"]), sd, enc, arch,
                                  synth.head_config(cfg["model"]), return_hidden=True)
    return g, cfg, labels, enc, arch, wav, logits, offs, hid


@pytest.mark.parametrize("name", list(GOLDEN_CASES))
def test_oracle_matches_reference(name, golden_dir):
    g, cfg, labels, enc, arch, wav, logits, offs, hid = _run(name, golden_dir)
    if "logits" in g:
        np.testing.assert_allclose(hid.numpy(), g["hidden"], atol=2e-5, rtol=0)
        np.testing.assert_allclose(logits.numpy(), g["logits"], atol=1e-4, rtol=0)
        if "logmel" in g:
            fe = O.whisper_log_mel(wav, arch.n_mels, arch.max_positions * 2 * arch.hop)
            np.testing.assert_allclose(fe.numpy(), g["logmel"], atol=1e-6, rtol=0)
    else:
        r = g["rows"]
        np.testing.assert_allclose(hid.numpy()[:, r], g["hidden_rows"], atol=2e-5, rtol=0)
        np.testing.assert_allclose(logits.numpy()[:, r], g["logits_rows"], atol=1e-4, rtol=0)
        fe = O.whisper_log_mel(wav, arch.n_mels, arch.max_positions * 2 * arch.hop)
        np.testing.assert_allclose(fe.numpy()[:, :, g["logmel_frames"]], g["logmel_cols"], atol=1e-6, rtol=0)
    np.testing.assert_allclose(offs.numpy(), g["offsets"], atol=1e-5, rtol=0)
    ids, maxp, arg, margin = O.tags_from_logits(logits, labels.index("O"), 0.5)
    np.testing.assert_allclose(maxp.numpy(), g["maxprob"], atol=2e-5, rtol=0)
    # ids must agree on every frame whose reference top-2 margin exceeds fp32 noise
    safe = g["margin"] > 1e-3
    assert (arg.numpy()[safe] == g["argmax"][safe]).all()
    assert safe.mean() > 0.99


def test_mel_filter_bank_properties():
    fb = O.mel_filter_bank(80)
    assert fb.shape == (201, 80) and (fb >= 0).all()
    # Slaney triangles: every filter non-empty, each bin feeds at most two filters
    assert (fb.max(axis=0) > 0).all()
    assert ((fb > 0).sum(axis=1) <= 2).all()
    assert O.mel_filter_bank(128).shape == (201, 128)


def test_whisper_log_mel_pads_and_truncates():
    x = torch.from_numpy(synth.make_batch(0, 1, 16000))
    a = O.whisper_log_mel(x, 80, 32000)
    b = O.whisper_log_mel(torch.nn.functional.pad(x, (0, 16000)), 80, 32000)
    c = O.whisper_log_mel(torch.nn.functional.pad(x, (0, 40000)), 80, 32000)
    assert a.shape == (1, 80, 200)
    assert torch.equal(a, b) and torch.equal(a, c)

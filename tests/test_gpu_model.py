"""-m gpu: the whole HIP forward (through wfl_forward) against the golden fixtures (outputs of the reference) and
against the oracle on the same seeded inputs, plus size-independent properties at BASELINE config-2 size.

Tolerances (bf16 MFMA operands with fp32 accumulation; residual stream carried as bf16 hi + lo, classifier in split
precision), stated once here.  tests/study_quant.py shows where the error comes from on the cfg2 fixture: rounding the WEIGHTS
to bf16 alone moves the logits by 0.155 max / 0.033 mean (a fixed property of a bf16-weight deployment), activation rounding by
0.076 / 0.014.  So there are two targets:

  (A) the reference on the checkpoint as given (fp32 weights) -- every fixture except *_bf16w:
      log-mel (fp32 path)        |err| <= 2e-3 everywhere, mean |err| <= 1e-4       (near-floor bins are fp32-FFT noise)
      encoder hidden (LN output) |err| <= 0.05 abs (values are O(1)), mean |err| <= 0.008
      logits (std ~6.5)          |err| <= 0.40 abs, mean |err| <= 0.07
      max-prob                   |err| <= 0.08;  offsets |err| <= 0.02
      tag ids                    identical on every frame whose fp32 top-2 logit margin > TAU = 0.4 and whose max-prob is
                                 further than BAND = 0.06 from the threshold
  (B) the reference on the bf16-rounded checkpoint (synth.round_weights_bf16; fixture whisper_base_cfg2_bf16w, and the oracle on
      the rounded weights in the held-out test) -- what is left is activation rounding:
      logits |err| <= 0.15 abs, mean <= 0.03;  max-prob <= 0.04;  tag ids identical wherever margin > TAU_W = 0.2 and
      |max-prob - threshold| > BAND_W = 0.03, which must cover >= 80 % of the fixture's frames (measured 83 %); raw argmax
      mismatches < 1 % (measured 0.87 %: all of them near-ties, 7.9 % of the fixture's frames have a top-2 margin <= 0.2).
  Models whose logits live on another scale (cfg3's are ~20x smaller) scale TAU with the reference logits' standard deviation:
  tau = TAU * std / 6.5 (6.5 = the cfg2 fixture's).
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import wfl_oracle as O
import synthetic as synth
from wfl_asr_amd.archs import resolve_encoder_arch
from wfl_asr_amd.tagger import BIOPhonemeTagger
from cases import GOLDEN_CASES, tiny_whisper_config

pytestmark = pytest.mark.gpu

TAU, BAND = 0.4, 0.06
TAU_W, BAND_W = 0.2, 0.03
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")


def _note(name, **kw):
    try:
        os.makedirs(OUT, exist_ok=True)
        with open(os.path.join(OUT, "parity_stats.jsonl"), "a") as f:
            f.write(json.dumps(dict(test=name, **{k: (float(v) if hasattr(v, "__float__") else v) for k, v in kw.items()})) + "\n")
    except OSError:
        pass


def _build(cfg, n_phonemes, seed):
    labels = synth.make_labels(n_phonemes)
    sd_np = synth.make_state_dict(cfg, len(labels), seed=seed)
    m = BIOPhonemeTagger(cfg, labels)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
    m.to("cuda").eval()
    return m, labels, sd_np


def _oracle(cfg, labels, sd_np, wav, lang, act_fp8=False):
    enc, arch = resolve_encoder_arch(cfg["model"])
    sd = O.to_torch_state_dict(sd_np)
    lg, of, hid = O.forward(torch.from_numpy(wav), None if lang is None else torch.from_numpy(lang), sd, enc, arch,
                            synth.head_config(cfg["model"]), return_hidden=True, act_fp8=act_fp8)
    return lg, of, hid


def _check_decisions(name, out, ref_logits, ref_offsets, o_id, thr):
    ids_ref, maxp_ref, arg_ref, margin = O.tags_from_logits(ref_logits, o_id, thr)
    lg = out.logits.cpu()
    err = (lg - ref_logits).abs()
    mp_err = (out.maxprob.cpu() - maxp_ref).abs()
    of_err = (out.offsets.cpu() - ref_offsets).abs()
    safe = (margin > TAU) & ((maxp_ref - thr).abs() > BAND)
    arg_bad = int((out.argmax.cpu().long() != arg_ref)[margin > TAU].sum())
    ids_bad = int((out.ids.cpu().long() != ids_ref)[safe].sum())
    _note(name, logits_max=err.max(), logits_mean=err.mean(), maxprob_max=mp_err.max(), offsets_max=of_err.max(),
          safe_frac=safe.float().mean(), argmax_bad=arg_bad, ids_bad=ids_bad,
          argmax_all_mismatch=int((out.argmax.cpu().long() != arg_ref).sum()), frames=int(arg_ref.numel()))
    assert err.max() <= 0.40 and err.mean() <= 0.07, (err.max(), err.mean())
    assert mp_err.max() <= 0.08 and of_err.max() <= 0.02, (mp_err.max(), of_err.max())
    assert arg_bad == 0 and ids_bad == 0
    assert safe.float().mean() >= 0.60


def test_logmel_matches_oracle():
    cfg = synth.baseline_config(1)
    m, labels, _ = _build(cfg, 70, seed=1)
    wav = np.zeros((3, 480000), np.float32)
    wav[0] = synth.make_clip(1000, 480000, seed=1)
    wav[1, :160000] = synth.make_clip(1001, 160000, seed=1)       # short clip, zero padded
    wav[2] = 0.0                                                  # silence: every bin at the 1e-10 clamp
    got = m.log_mel(torch.from_numpy(wav).cuda()).cpu()
    ref = O.whisper_log_mel(torch.from_numpy(wav), 80, 480000)
    err = (got - ref).abs()
    _note("logmel", max=err.max(), mean=err.mean())
    assert err.max() <= 2e-3 and err.mean() <= 1e-4
    assert float((got[2] + 1.5).abs().max()) <= 1e-6               # silence: constant -1.5 everywhere
    # lens: a ragged batch equals explicit zero padding
    lens = torch.tensor([480000, 160000, 0], dtype=torch.int32)
    junk = wav.copy(); junk[1, 160000:] = 0.37; junk[2] = -0.2
    got2 = m.log_mel(torch.from_numpy(junk).cuda(), lens=lens).cpu()
    assert torch.equal(got, got2)


@pytest.mark.parametrize("name", ["whisper_base_cfg2", "whisper_base_cfg2_bf16w"])
def test_forward_matches_reference_golden(name, golden_dir):
    """BASELINE config 2 against outputs of the reference itself: on the checkpoint as given (target A) and on the bf16-rounded
    checkpoint (target B, tight)."""
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    cfg = GOLDEN_CASES[name]()
    bf16w = "bf16_weights" in g and bool(int(g["bf16_weights"]))
    labels = synth.make_labels(int(g["n_phonemes"]))
    sd_np = synth.make_state_dict(cfg, len(labels), seed=int(g["seed"]))
    if bf16w:
        sd_np = synth.round_weights_bf16(sd_np)
    m = BIOPhonemeTagger(cfg, labels)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
    m.to("cuda").eval()
    B, L = len(g["lang_id"]), int(g["L"])
    wav = synth.make_batch(int(g["clip0"]), B, L, seed=int(g["seed"]))
    thr = 0.5
    out = m.label(torch.from_numpy(wav).cuda(), g["lang_id"], threshold=thr, want_logits=True, want_hidden=True)
    torch.cuda.synchronize()
    r = g["rows"]
    hid_err = np.abs(out.hidden.cpu().numpy()[:, r] - g["hidden_rows"])
    lg_err = np.abs(out.logits.cpu().numpy()[:, r] - g["logits_rows"])
    mp_err = np.abs(out.maxprob.cpu().numpy() - g["maxprob"])
    of_err = np.abs(out.offsets.cpu().numpy() - g["offsets"])
    o_id = labels.index("O")
    ids_ref = np.where(g["maxprob"] < thr, o_id, g["argmax"].astype(np.int64))
    tau, band = (TAU_W, BAND_W) if bf16w else (TAU, BAND)
    safe = (g["margin"] > tau) & (np.abs(g["maxprob"] - thr) > band)
    arg_bad = int((out.argmax.cpu().numpy() != g["argmax"])[g["margin"] > tau].sum())
    ids_bad = int((out.ids.cpu().numpy() != ids_ref)[safe].sum())
    raw = int((out.argmax.cpu().numpy() != g["argmax"]).sum())
    _note("golden_" + name, hidden_max=hid_err.max(), hidden_mean=hid_err.mean(), logits_max=lg_err.max(),
          logits_mean=lg_err.mean(), maxprob_max=mp_err.max(), offsets_max=of_err.max(), tau=tau, band=band, safe_frac=safe.mean(),
          argmax_bad=arg_bad, ids_bad=ids_bad, argmax_all_mismatch=raw, frames=int(g["argmax"].size))
    assert arg_bad == 0 and ids_bad == 0
    if bf16w:
        assert hid_err.max() <= 0.04 and hid_err.mean() <= 0.005
        assert lg_err.max() <= 0.15 and lg_err.mean() <= 0.03
        assert mp_err.max() <= 0.04 and of_err.max() <= 0.01
        assert safe.mean() >= 0.80 and raw < 0.01 * g["argmax"].size
    else:
        assert hid_err.max() <= 0.05 and hid_err.mean() <= 0.008
        assert lg_err.max() <= 0.40 and lg_err.mean() <= 0.07
        assert mp_err.max() <= 0.08 and of_err.max() <= 0.02
        assert safe.mean() >= 0.65


def _tiny(**kw):
    return tiny_whisper_config(enable_bilstm=False, **kw)


def test_tiny_forward_vs_oracle_ragged_batch():
    cfg = _tiny()
    m, labels, sd_np = _build(cfg, 5, seed=21)
    B, L = 3, 32000
    lens = np.array([32000, 17000, 4000], np.int32)
    wav = np.zeros((B, L), np.float32)
    for i, n in enumerate(lens):
        wav[i, :n] = synth.make_clip(300 + i, int(n), seed=21)
    lang = np.array([0, 1, 0], np.int64)
    out = m.label(torch.from_numpy(wav).cuda(), lang, threshold=0.5, lens=lens, want_logits=True, want_hidden=True)
    torch.cuda.synchronize()
    lg, of, hid = _oracle(cfg, labels, sd_np, wav, lang)
    h_err = (out.hidden.cpu() - hid).abs()
    _note("tiny_hidden", max=h_err.max(), mean=h_err.mean())
    assert h_err.max() <= 0.05
    _check_decisions("tiny_ragged", out, lg, of, labels.index("O"), 0.5)


def test_language_modes_vs_oracle():
    cfg = _tiny(num_languages=3)
    m, labels, sd_np = _build(cfg, 5, seed=22)
    wav = synth.make_batch(400, 2, 20000, seed=22)
    x = torch.from_numpy(wav).cuda()
    # forward(lang_id=None): lang_proj skipped (model.py:176)
    out = m.label(x, None, threshold=0.3, want_logits=True)
    lg, of, _ = _oracle(cfg, labels, sd_np, wav, None)
    _check_decisions("lang_none", out, lg, of, labels.index("O"), 0.3)
    # infer.py:266-276: mean over every language id
    out = m.label(x, None, threshold=0.3, average_languages=True, want_logits=True)
    lgs, ofs = [], []
    for lid in range(3):
        a, b, _ = _oracle(cfg, labels, sd_np, wav, np.full(2, lid, np.int64))
        lgs.append(a); ofs.append(b)
    _check_decisions("lang_avg", out, torch.stack(lgs).mean(0), torch.stack(ofs).mean(0), labels.index("O"), 0.3)
    with pytest.raises(ValueError):
        m.label(x, [0, 3])


def test_dilated_head_vs_oracle():
    cfg = tiny_whisper_config(enable_bilstm=False, num_conformer_layers=1, dilated_conv_depth=3, dilated_conv_kernel=5)
    m, labels, sd_np = _build(cfg, 5, seed=23)
    wav = synth.make_batch(500, 2, 32000, seed=23)
    lang = np.array([1, 0], np.int64)
    out = m.label(torch.from_numpy(wav).cuda(), lang, threshold=0.5, want_logits=True)
    lg, of, _ = _oracle(cfg, labels, sd_np, wav, lang)
    _check_decisions("dilated", out, lg, of, labels.index("O"), 0.5)


def test_full_size_properties_cfg2():
    """BASELINE config 2 at its full size (16 x 30 s): properties that need no oracle run."""
    cfg = synth.baseline_config(1)
    m, labels, _ = _build(cfg, 70, seed=1)
    B, L = 16, 480000
    wav = synth.make_batch(2000, B, L, seed=1)
    wav[5, 200000:] = 0.0
    lang = (np.arange(B) % 2).astype(np.int64)
    x = torch.from_numpy(wav).cuda()
    full = m.label(x, lang, threshold=0.5, want_logits=True)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(full.logits).all()) and bool(torch.isfinite(full.offsets).all())
    assert bool(((full.offsets >= 0) & (full.offsets <= 1)).all())
    assert bool(((full.maxprob > 0) & (full.maxprob <= 1.0 + 1e-6)).all())
    # decision rule holds frame by frame
    want = torch.where(full.maxprob < 0.5, torch.full_like(full.argmax, labels.index("O")), full.argmax)
    assert torch.equal(full.ids, want)
    assert torch.equal(full.argmax.long(), full.logits.argmax(-1))
    # batch invariance: a clip labelled alone / in a different batch slot gives bit-identical results
    for i in (0, 5, 15):
        one = m.label(x[i:i + 1], lang[i:i + 1], threshold=0.5, want_logits=True)
        assert torch.equal(one.logits[0], full.logits[i]), i
        assert torch.equal(one.offsets[0], full.offsets[i]) and torch.equal(one.ids[0], full.ids[i])
    # Whisper pads to 30 s itself: a truncated clip == the same clip zero-padded (clip 5 is silent after 12.5 s)
    short = m.label(x[5:6, :200000].contiguous(), lang[5:6], threshold=0.5, want_logits=True)
    assert torch.equal(short.logits[0], full.logits[5])
    # determinism
    again = m.label(x, lang, threshold=0.5, want_logits=True)
    assert torch.equal(again.logits, full.logits) and torch.equal(again.ids, full.ids)
    _note("cfg2_full", distinct_tags=int(full.ids.unique().numel()), o_frac=float((full.ids == labels.index("O")).float().mean()))


@pytest.mark.parametrize("units", [None, "8"])
def test_bilstm_tiny_matches_reference_golden(units, golden_dir, monkeypatch):
    """whisper_tiny golden = the reference's full default head (2-layer BiLSTM, 2 Conformer, 2 dilated convs), full
    tensors.  units=8 forces 4 workgroups per direction so the inter-workgroup hand-off is exercised at H=32."""
    if units:
        monkeypatch.setenv("WFL_LSTM_UNITS", units)
    g = np.load(os.path.join(golden_dir, "whisper_tiny.npz"))
    cfg = GOLDEN_CASES["whisper_tiny"]()
    m, labels, sd_np = _build(cfg, int(g["n_phonemes"]), int(g["seed"]))
    B, L = len(g["lang_id"]), int(g["L"])
    wav = synth.make_batch(int(g["clip0"]), B, L, seed=int(g["seed"]))
    out = m.label(torch.from_numpy(wav).cuda(), g["lang_id"], threshold=0.5, want_logits=True, want_hidden=True)
    m.check(B, L)
    h_err = np.abs(out.hidden.cpu().numpy() - g["hidden"])
    assert h_err.max() <= 0.08
    _check_decisions("tiny_bilstm_" + str(units), out, torch.from_numpy(g["logits"]), torch.from_numpy(g["offsets"]),
                     labels.index("O"), 0.5)
    again = m.label(torch.from_numpy(wav).cuda(), g["lang_id"], threshold=0.5, want_logits=True)
    assert torch.equal(again.logits, out.logits)


def test_bilstm_base_matches_reference_golden(golden_dir):
    """Whisper-base + the default config.yaml head (H=256: 4 workgroups per direction), B=1, 18.75 s clip."""
    g = np.load(os.path.join(golden_dir, "whisper_base_full.npz"))
    cfg = GOLDEN_CASES["whisper_base_full"]()
    m, labels, sd_np = _build(cfg, int(g["n_phonemes"]), int(g["seed"]))
    B, L = len(g["lang_id"]), int(g["L"])
    wav = synth.make_batch(int(g["clip0"]), B, L, seed=int(g["seed"]))
    thr = 0.5
    out = m.label(torch.from_numpy(wav).cuda(), g["lang_id"], threshold=thr, want_logits=True)
    m.check(B, L)
    r = g["rows"]
    lg_err = np.abs(out.logits.cpu().numpy()[:, r] - g["logits_rows"])
    mp_err = np.abs(out.maxprob.cpu().numpy() - g["maxprob"])
    of_err = np.abs(out.offsets.cpu().numpy() - g["offsets"])
    safe = (g["margin"] > TAU) & (np.abs(g["maxprob"] - thr) > BAND)
    o_id = labels.index("O")
    ids_ref = np.where(g["maxprob"] < thr, o_id, g["argmax"].astype(np.int64))
    arg_bad = int((out.argmax.cpu().numpy() != g["argmax"])[g["margin"] > TAU].sum())
    ids_bad = int((out.ids.cpu().numpy() != ids_ref)[safe].sum())
    _note("golden_base_full", logits_max=lg_err.max(), logits_mean=lg_err.mean(), maxprob_max=mp_err.max(),
          offsets_max=of_err.max(), safe_frac=safe.mean(), argmax_bad=arg_bad, ids_bad=ids_bad)
    assert lg_err.max() <= 0.40 and lg_err.mean() <= 0.07
    assert mp_err.max() <= 0.08 and of_err.max() <= 0.02
    assert arg_bad == 0 and ids_bad == 0 and safe.mean() >= 0.60
    # batch of 20 clips = 2 clip groups (one partial): clip 0 is bit-identical to the B=1 run
    wav20 = np.concatenate([wav, synth.make_batch(7000, 19, L, seed=3)])
    lang20 = np.concatenate([g["lang_id"], np.arange(19) % 2]).astype(np.int64)
    big = m.label(torch.from_numpy(wav20).cuda(), lang20, threshold=thr, want_logits=True)
    m.check(20, L)
    assert torch.equal(big.logits[0], out.logits[0])
    assert bool(torch.isfinite(big.logits).all())


@pytest.mark.parametrize("name", ["wavlm_tiny_group", "wavlm_tiny_stable", "wavlm_base_cfg1"])
def test_wavlm_matches_reference_golden(name, golden_dir):
    """WavLM encoder (group-norm / post-LN and layer-norm / stable-pre-LN topologies; BASELINE config 1 = WavLM-base
    dims + linear head on a 1 s 440 Hz sine) against full tensors recorded from the reference."""
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    cfg = GOLDEN_CASES[name]()
    m, labels, sd_np = _build(cfg, int(g["n_phonemes"]), int(g["seed"]))
    B, L = len(g["lang_id"]), int(g["L"])
    wav = np.stack([synth.sine_clip(L)] * B) if name == "wavlm_base_cfg1" else synth.make_batch(int(g["clip0"]), B, L, seed=int(g["seed"]))
    out = m.label(torch.from_numpy(wav).cuda(), g["lang_id"], threshold=0.5, want_logits=True, want_hidden=True)
    m.check(B, L)
    assert tuple(out.logits.shape) == g["logits"].shape
    h_err = np.abs(out.hidden.cpu().numpy() - g["hidden"])
    _note("wavlm_hidden_" + name, max=h_err.max(), mean=h_err.mean(), ref_abs_mean=np.abs(g["hidden"]).mean())
    assert h_err.max() <= 0.08 and h_err.mean() <= 0.012
    _check_decisions(name, out, torch.from_numpy(g["logits"]), torch.from_numpy(g["offsets"]), labels.index("O"), 0.5)
    again = m.label(torch.from_numpy(wav).cuda(), g["lang_id"], threshold=0.5, want_logits=True)
    assert torch.equal(again.logits, out.logits)


@pytest.mark.parametrize("idx,B,L", [(3, 1, 160000), (2, 2, 48000)])
def test_baseline_configs_3_and_4_vs_oracle(idx, B, L):
    """BASELINE configs[3] (Whisper-small + full default head: BiLSTM H=384, Conformer head_dim 384) and configs[2]
    (WavLM-large, layer-norm feature encoder + stable layer norm, BiLSTM H=512 + dilated stack) at their real widths,
    on short clips so the CPU oracle finishes in seconds."""
    cfg = synth.baseline_config(idx)
    m, labels, sd_np = _build(cfg, 70, seed=40 + idx)
    wav = synth.make_batch(900 + idx, B, L, seed=40 + idx)
    lang = (np.arange(B) % 2).astype(np.int64)
    out = m.label(torch.from_numpy(wav).cuda(), lang, threshold=0.5, want_logits=True, want_hidden=True)
    m.check(B, L)
    lg, of, hid = _oracle(cfg, labels, sd_np, wav, lang)
    h_err = (out.hidden.cpu() - hid).abs()
    _note(f"cfg{idx + 1}_hidden", max=h_err.max(), mean=h_err.mean())
    assert h_err.max() <= 0.08 and h_err.mean() <= 0.012
    ids_ref, maxp_ref, arg_ref, margin = O.tags_from_logits(lg, labels.index("O"), 0.5)
    err = (out.logits.cpu() - lg).abs()
    tau = TAU * float(lg.std()) / 6.5
    safe = margin > tau
    bad = int((out.argmax.cpu().long() != arg_ref)[safe].sum())
    _note(f"cfg{idx + 1}", logit_std=lg.std(), tau=tau, logits_max=err.max(), logits_mean=err.mean(), offsets_max=(out.offsets.cpu() - of).abs().max(),
          safe_frac=safe.float().mean(), argmax_bad=bad, argmax_all_mismatch=int((out.argmax.cpu().long() != arg_ref).sum()),
          frames=int(arg_ref.numel()))
    assert err.max() <= 0.40 and err.mean() <= 0.08
    assert (out.offsets.cpu() - of).abs().max() <= 0.02
    assert bad == 0 and safe.float().mean() >= 0.5


def _cfg5_case(case, activation_dtype=None):
    cfg = synth.baseline_config(4)
    assert cfg["model"]["weight_dtype"] == "fp8"
    if case == "large_v3_4l":
        cfg["model"]["whisper_model"] = "local/whisper-large-v3-4l"
        cfg["model"]["encoder_arch"] = dict(d_model=1280, layers=4, heads=20, ffn=5120, n_mels=128, max_positions=1500)
    else:
        cfg["model"]["whisper_model"] = "local/whisper-base-fp8"
        cfg["model"]["encoder_arch"] = dict(d_model=512, layers=6, heads=8, ffn=2048, n_mels=80, max_positions=1500)
    if activation_dtype:
        cfg["model"]["activation_dtype"] = activation_dtype
    return cfg


# What fp8 WEIGHTS cost inside the reference's own arithmetic (oracle on the fp8-rounded checkpoint vs oracle on the checkpoint as
# given, computed on CPU with no build involved; tests/study_fp8.py): logits 1.51 / 0.315 (large_v3_4l), 1.41 / 0.265 (base_6l); 73 / 183
# of 1500 raw argmax decisions differ, 18 / 37 of them at a margin above 1 tau, 5 / 4 above 2 tau, none above 4 tau (55 % / 30 % of the
# frames).  That is the price of BASELINE configs[4]'s "fp8 weights" on these synthetic checkpoints and no kernel can change it: the
# F32 bounds below are that distance plus the build's own (bf16-sized) share, with the tag rule at 4 tau.
@pytest.mark.parametrize("case", ["large_v3_4l", "base_6l"])
def test_baseline_config_5_fp8_weights_vs_oracle(case):
    """BASELINE configs[4]: Whisper-large-v3 encoder with fp8 weights (model.weight_dtype: fp8), linear head, DEFAULT build = e4m3 weights
    (one scale per output channel), bf16 activations.  "large_v3_4l" is the real geometry (128 mel bins, d = 1280, 20 heads, FFN 5120)
    with 4 of the 32 layers so that the CPU oracle finishes in seconds (tests/test_gpu_round2.py holds one row of the full 32-layer,
    32-clip forward to the oracle too); "base_6l" runs ALL layers of a smaller fp8 encoder (Whisper-base dims).
    Targets, both outputs of the oracle = the reference's arithmetic, with FIXED bounds (round 4: nothing here is derived from the build):
      W8  (primary) the reference on the fp8-rounded checkpoint, exact activations: the standard bf16 tolerances and the 1-tau tag rule;
      F32 the reference on the checkpoint as given: the format's own distance (note above) + the build's share, tag rule at 4 tau."""
    cfg = _cfg5_case(case)
    m, labels, sd_np = _build(cfg, 70, seed=45)
    B, L = 1, 160000
    wav = synth.make_batch(905, B, L, seed=45)
    lang = np.zeros(B, np.int64)
    out = m.label(torch.from_numpy(wav).cuda(), lang, threshold=0.5, want_logits=True, want_hidden=True)
    m.check(B, L)
    sd8 = synth.round_weights_fp8(sd_np)
    for target, sd_t, tmul in (("W8", sd8, 1.0), ("F32", sd_np, 4.0)):
        lg, of, hid = _oracle(cfg, labels, sd_t, wav, lang)
        h_err = (out.hidden.cpu() - hid).abs()
        ids_ref, maxp_ref, arg_ref, margin = O.tags_from_logits(lg, labels.index("O"), 0.5)
        err = (out.logits.cpu() - lg).abs()
        of_err = (out.offsets.cpu() - of).abs().max()
        tau = tmul * TAU * float(lg.std()) / 6.5
        safe = margin > tau
        mism = out.argmax.cpu().long() != arg_ref
        bad, raw = int(mism[safe].sum()), float(mism.float().mean())
        _note(f"cfg5_fp8_{case}_{target}", hidden_max=h_err.max(), hidden_mean=h_err.mean(), logit_std=lg.std(), tau=tau,
              logits_max=err.max(), logits_mean=err.mean(), offsets_max=of_err, safe_frac=safe.float().mean(), argmax_bad=bad,
              raw_mismatch_rate=raw, frames=int(arg_ref.numel()))
        if target == "W8":
            assert h_err.max() <= 0.08 and h_err.mean() <= 0.012, (float(h_err.max()), float(h_err.mean()))
            assert err.max() <= 0.40 and err.mean() <= 0.08, (float(err.max()), float(err.mean()))
            assert of_err <= 0.02
            assert raw <= 0.02, raw                                 # (the oracle with bf16-rounded GEMM inputs: 0.4 %)
            assert bad == 0 and safe.float().mean() >= 0.7, (bad, float(safe.float().mean()))
        else:
            assert err.max() <= 1.9 and err.mean() <= 0.40, (float(err.max()), float(err.mean()))      # format: 1.51 / 0.315, 1.41 / 0.265
            assert of_err <= 0.08
            assert raw <= 0.15, raw                                 # format: 4.9 % / 12.2 %
            assert bad == 0 and safe.float().mean() >= 0.25, (bad, float(safe.float().mean()))         # graded at 4 tau: 55 % / 30 %


@pytest.mark.parametrize("case", ["large_v3_4l", "base_6l"])
def test_fp8_pair_activations_hold_the_reference_on_the_fp8_checkpoint(case):
    """`model.activation_dtype: fp8_pair` (round 4, csrc/gemm_mx.hip): every GEMM input of an encoder layer as an e4m3 PAIR hi + lo on the
    block-scaled fp8 MFMA (v_mfma_scale_f32_16x16x128_f8f6f4; lo rides in the same instruction with a 2^-4 block scale) -- eight
    significant bits per activation, like bf16.  Held to exactly the bounds of the default build (bf16 activations) against W8, the
    reference on the fp8-rounded checkpoint; tests/study_fp8.py: the reference's own arithmetic with pair-rounded GEMM inputs sits
    0.06 / 0.009 from itself."""
    cfg = _cfg5_case(case, "fp8_pair")
    m, labels, sd_np = _build(cfg, 70, seed=45)
    B, L = 1, 160000
    wav = synth.make_batch(905, B, L, seed=45)
    lang = np.zeros(B, np.int64)
    out = m.label(torch.from_numpy(wav).cuda(), lang, threshold=0.5, want_logits=True, want_hidden=True)
    m.check(B, L)
    assert int(out.status.item()) == 0
    lg, of, hid = _oracle(cfg, labels, synth.round_weights_fp8(sd_np), wav, lang)
    h_err = (out.hidden.cpu() - hid).abs()
    ids_ref, maxp_ref, arg_ref, margin = O.tags_from_logits(lg, labels.index("O"), 0.5)
    err = (out.logits.cpu() - lg).abs()
    safe = margin > TAU * float(lg.std()) / 6.5
    mism = out.argmax.cpu().long() != arg_ref
    _note(f"cfg5_fp8pair_{case}_W8", hidden_max=h_err.max(), hidden_mean=h_err.mean(), logits_max=err.max(), logits_mean=err.mean(),
          offsets_max=(out.offsets.cpu() - of).abs().max(), safe_frac=safe.float().mean(), argmax_bad=int(mism[safe].sum()),
          raw_mismatch_rate=mism.float().mean())
    assert h_err.max() <= 0.08 and h_err.mean() <= 0.012, (float(h_err.max()), float(h_err.mean()))
    assert err.max() <= 0.40 and err.mean() <= 0.08, (float(err.max()), float(err.mean()))
    assert (out.offsets.cpu() - of).abs().max() <= 0.02
    assert float(mism.float().mean()) <= 0.02
    assert int(mism[safe].sum()) == 0 and safe.float().mean() >= 0.7
    # a clip labelled alone equals the clip inside a batch, bit for bit (tiles straddle clips; the pair planes' halo rows are never read into a stored row)
    wav3 = np.concatenate([synth.make_batch(906, 2, L, seed=45), wav])
    out3 = m.label(torch.from_numpy(wav3).cuda(), np.zeros(3, np.int64), threshold=0.5, want_logits=True)
    assert torch.equal(out3.logits[2], out.logits[0])


@pytest.mark.parametrize("case,act", [("large_v3_4l", "fp8"), ("base_6l", "fp8"), ("base_6l", "fp8_nonscaled")])
def test_fp8_activations_are_an_opt_in_with_a_stated_price(case, act):
    """`model.activation_dtype: fp8` (wfl_arch::fp8_activations; round 3's default, an opt-in since round 4): the four GEMM inputs of
    every encoder layer are e4m3 too and the GEMMs run fp8 x fp8 -- on the block-scaled MFMA ("fp8", round 4: csrc/gemm_mx.hip) or on
    round 3's non-scaled one ("fp8_nonscaled").  Three mantissa bits on the activations are not parity with the
    reference -- tests/study_fp8.py: the reference's own arithmetic with e4m3 GEMM inputs sits 1.1 / 0.19 (logits max / mean) from
    itself with exact ones, 82 of 1500 raw argmax decisions differ -- so this mode is held to what it is sold as:
      * against W8 (the reference on the fp8 checkpoint): logits mean <= 0.30, raw tag mismatch <= 10 %, none above 4 tau;
      * additionally, against the oracle with the same rounding points (A8, a diagnostic of the format, NOT the reference): the build is
        not further from it than 1.25 x the format's own A8 - W8 distance, i.e. it adds nothing of its own;
      * a clean status word (bit 1 = an activation saturated at its fixed scale)."""
    cfg = _cfg5_case(case, act)
    m, labels, sd_np = _build(cfg, 70, seed=45)
    B, L = 1, 160000
    wav = synth.make_batch(905, B, L, seed=45)
    lang = np.zeros(B, np.int64)
    out = m.label(torch.from_numpy(wav).cuda(), lang, threshold=0.5, want_logits=True, want_hidden=True)
    m.check(B, L)
    assert int(out.status.item()) == 0
    sd8 = synth.round_weights_fp8(sd_np)
    w8 = _oracle(cfg, labels, sd8, wav, lang)
    a8 = _oracle(cfg, labels, sd8, wav, lang, act_fp8=True)
    fmt_l = (a8[0] - w8[0]).abs()
    lg, of, hid = w8
    ids_ref, maxp_ref, arg_ref, margin = O.tags_from_logits(lg, labels.index("O"), 0.5)
    err = (out.logits.cpu() - lg).abs()
    mism = out.argmax.cpu().long() != arg_ref
    safe = margin > 4.0 * TAU * float(lg.std()) / 6.5
    err_a8 = (out.logits.cpu() - a8[0]).abs()
    _note(f"cfg5_fp8act_{case}_{act}", logits_max=err.max(), logits_mean=err.mean(), raw_mismatch_rate=mism.float().mean(),
          safe_frac=safe.float().mean(), argmax_bad=int(mism[safe].sum()), format_logits_mean=fmt_l.mean(), format_logits_max=fmt_l.max(),
          vs_a8_mean=err_a8.mean(), vs_a8_max=err_a8.max())
    assert err.mean() <= 0.30 and err.max() <= 2.0, (float(err.max()), float(err.mean()))
    assert float(mism.float().mean()) <= 0.10
    assert int(mism[safe].sum()) == 0 and safe.float().mean() >= 0.2
    assert (out.offsets.cpu() - of).abs().max() <= 0.08
    assert err_a8.mean() <= 1.25 * fmt_l.mean() and err_a8.max() <= 1.5 * fmt_l.max(), (float(err_a8.mean()), float(fmt_l.mean()))


def test_fp8_activation_saturation_is_reported():
    """Advisor (round 3): the fixed scale 8 of the e4m3 attention context / GELU output clipped silently above 56.  A checkpoint whose
    fc1 bias pushes one GELU channel to ~ 70: the forward must set bit 1 of the status word, `check()` must raise, and the default
    (bf16 activations) build of the same checkpoint must run clean."""
    from wfl_asr_amd import _lib
    for act, want in (("fp8", 2), ("fp8_nonscaled", 2), ("fp8_pair", 0), (None, 0)):      # (70 fits a pair at scale 4: |x| <= 112)
        cfg = _cfg5_case("base_6l", act)
        labels = synth.make_labels(70)
        sd_np = synth.make_state_dict(cfg, len(labels), seed=46)
        b = sd_np["encoder.layers.2.fc1.bias"].copy()
        b[5] = 70.0
        sd_np["encoder.layers.2.fc1.bias"] = b
        m = BIOPhonemeTagger(cfg, labels)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
        m.to("cuda").eval()
        wav = synth.make_batch(906, 1, 48000, seed=46)
        out = m.label(torch.from_numpy(wav).cuda(), np.zeros(1, np.int64), threshold=0.5)
        assert int(out.status.item()) & 2 == want, (act, int(out.status.item()))
        if want:
            with pytest.raises(_lib.WflError):
                m.check(1, 48000)


def test_graph_replay_is_bit_identical():
    cfg = _tiny()
    m, labels, _ = _build(cfg, 5, seed=24)
    wav = torch.from_numpy(synth.make_batch(600, 4, 32000, seed=24)).cuda()
    lang = np.array([0, 1, 1, 0], np.int64)
    eager = m.label(wav, lang, threshold=0.5, want_logits=True)
    for rep in range(3):
        w = torch.roll(wav, rep, 0)
        la = np.roll(lang, rep)
        g = m.label(w, la, threshold=0.5, want_logits=True, graph=True)
        e = m.label(w, la, threshold=0.5, want_logits=True)
        assert torch.equal(g.logits, e.logits) and torch.equal(g.ids, e.ids) and torch.equal(g.offsets, e.offsets)
    assert torch.equal(m.label(wav, lang, threshold=0.5, want_logits=True, graph=True).logits, eager.logits)


def test_errors_are_loud():
    cfg = _tiny()
    labels = synth.make_labels(5)
    m = BIOPhonemeTagger(cfg, labels)
    sd = {k: torch.from_numpy(v) for k, v in synth.make_state_dict(cfg, len(labels), seed=1).items()}
    bad = dict(sd); bad.pop("classifier.bias")
    with pytest.raises(RuntimeError, match="missing key"):
        BIOPhonemeTagger(cfg, labels).load_state_dict(bad)
    bad = dict(sd); bad["extra.weight"] = torch.zeros(3)
    with pytest.raises(RuntimeError, match="unexpected key"):
        BIOPhonemeTagger(cfg, labels).load_state_dict(bad)
    bad = dict(sd); bad["classifier.weight"] = torch.zeros(3, 64)
    with pytest.raises(RuntimeError, match="size mismatch"):
        BIOPhonemeTagger(cfg, labels).load_state_dict(bad)
    with pytest.raises(RuntimeError):
        m.label(torch.zeros(1, 100).cuda())                      # not loaded
    m.load_state_dict(sd)
    with pytest.raises(RuntimeError):
        m.label(torch.zeros(1, 100))                             # CPU tensor: no CPU path

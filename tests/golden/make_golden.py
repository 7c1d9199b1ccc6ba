#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE in the build container.

Dev-only: needs /root/reference and `transformers` (neither exists on the GPU box, nothing in
tests/bench/smoke imports this file).  Recipe from SURVEY.md §8c: import the HF classes first,
stub the absent `torchaudio` / `soundfile` modules (untouched on the Whisper/WavLM inference
path), put /root/reference on sys.path, build the reference's unmodified `BIOPhonemeTagger`
from a *local* HF directory (random init; no hub access), load the deterministic synthetic
state dict from `synthetic` strictly, and record its outputs on the synthetic clips.

Outputs (all small):
  whisper_tiny.npz        full tensors for a d=64 / 2-layer / 100-frame Whisper + full default head
  whisper_base_cfg2.npz   BASELINE config 2 (Whisper-base + 2 Conformer), B=2 x 30 s: ids, max-prob,
                          offsets, top-2 margin, sparse logits rows, log-mel/hidden samples
  whisper_base_cfg2_bf16w.npz   the same case with the checkpoint's weight tensors rounded to bf16 first (synth.round_weights_bf16)
  whisper_base_full.npz   Whisper-base + default config.yaml head (BiLSTM x2, Conformer x2, dilated x2)
  wavlm_*.npz             WavLM cases (BASELINE config 1 and a tiny stable-layer-norm variant)
  postprocess.json        outputs of the reference's own host functions on seeded inputs
  metrics.json            (round 4) train.py's compute_framewise_accuracy / compute_phoneme_error_rate / compute_timing_error / clean_lab
                          and correct_label.py's correct_lab_boundaries on seeded inputs.  Both modules import here with further EMPTY
                          stubs for the absent pytorch_optimizer / tensorboard / librosa (none is touched by these functions)
"""
import json
import os
import sys
import tempfile
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
os.environ.setdefault("HF_HUB_OFFLINE", "1")
os.environ.setdefault("TRANSFORMERS_OFFLINE", "1")
sys.dont_write_bytecode = True

import numpy as np
import torch

from transformers import (WhisperFeatureExtractor, WhisperModel, WhisperConfig, WavLMModel, WavLMConfig,
                          Wav2Vec2FeatureExtractor)

for _name in ("torchaudio", "torchaudio.transforms", "torchaudio.functional", "soundfile"):
    sys.modules.setdefault(_name, types.ModuleType(_name))
sys.modules["torchaudio"].transforms = sys.modules["torchaudio.transforms"]
sys.modules["torchaudio"].functional = sys.modules["torchaudio.functional"]

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

import model as ref_model      # noqa: E402  (the reference, unmodified)
import utils as ref_utils      # noqa: E402
import infer as ref_infer      # noqa: E402
from scipy.ndimage import median_filter  # noqa: E402

import synthetic as synth  # noqa: E402
from wfl_asr_amd.archs import resolve_encoder_arch  # noqa: E402

torch.manual_seed(0)
torch.set_grad_enabled(False)


def build_reference(config, labels, tmp):
    """Reference BIOPhonemeTagger over a locally saved random-init HF encoder of the right dims."""
    enc, arch = resolve_encoder_arch(config["model"])
    tag = getattr(arch, "feat_extract_norm", "")
    d = os.path.join(tmp, f"hf_{enc}_{arch.d_model}_{arch.layers}_{tag}")
    if not os.path.isdir(d):
        if enc == "whisper":
            hc = WhisperConfig(d_model=arch.d_model, encoder_layers=arch.layers, encoder_attention_heads=arch.heads,
                               encoder_ffn_dim=arch.ffn, num_mel_bins=arch.n_mels,
                               max_source_positions=arch.max_positions, decoder_layers=1,
                               decoder_attention_heads=arch.heads, decoder_ffn_dim=64, vocab_size=128,
                               pad_token_id=0, bos_token_id=1, eos_token_id=2, decoder_start_token_id=1,
                               suppress_tokens=None, begin_suppress_tokens=None)
            WhisperModel(hc).save_pretrained(d)
            WhisperFeatureExtractor(feature_size=arch.n_mels, hop_length=arch.hop, n_fft=arch.n_fft,
                                    chunk_length=arch.max_positions * 2 * arch.hop // 16000).save_pretrained(d)
        else:
            hc = WavLMConfig(hidden_size=arch.d_model, num_hidden_layers=arch.layers,
                             num_attention_heads=arch.heads, intermediate_size=arch.ffn,
                             conv_dim=list(arch.conv_dim), conv_kernel=list(arch.conv_kernel),
                             conv_stride=list(arch.conv_stride), feat_extract_norm=arch.feat_extract_norm,
                             conv_bias=arch.conv_bias, do_stable_layer_norm=arch.stable_layer_norm,
                             num_conv_pos_embeddings=arch.pos_conv_kernel,
                             num_conv_pos_embedding_groups=arch.pos_conv_groups,
                             num_buckets=arch.num_buckets, max_bucket_distance=arch.max_distance,
                             layer_norm_eps=arch.layer_norm_eps)
            WavLMModel(hc).save_pretrained(d)
            Wav2Vec2FeatureExtractor(do_normalize=arch.do_normalize, return_attention_mask=False).save_pretrained(d)
    cfg = json.loads(json.dumps(config))
    cfg["model"]["whisper_model" if enc == "whisper" else "wavlm_model"] = d
    m = ref_model.BIOPhonemeTagger(cfg, labels)
    return m, enc, arch


def load_synth(m, config, n_classes, seed, bf16_weights=False):
    sd_np = synth.make_state_dict(config, n_classes, seed=seed)
    if bf16_weights:                      # the reference, unmodified, on the checkpoint with bf16-rounded weight tensors
        sd_np = synth.round_weights_bf16(sd_np)
    ref_keys = set(m.state_dict().keys())
    mine = set(sd_np.keys())
    assert ref_keys == mine, (sorted(ref_keys - mine)[:10], sorted(mine - ref_keys)[:10])
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()}, strict=True)
    m.eval()
    return sd_np


def run_case(name, config, n_phonemes, B, L, seed, tmp, full=False, clip0=0, sine=False, rows=24, bf16_weights=False):
    labels = synth.make_labels(n_phonemes)
    m, enc, arch = build_reference(config, labels, tmp)
    load_synth(m, config, len(labels), seed, bf16_weights)
    if sine:
        wav = np.stack([synth.sine_clip(L)] * B)
    else:
        wav = synth.make_batch(clip0, B, L, seed=seed)
    lang = (np.arange(B) % config["model"]["num_languages"]).astype(np.int64)
    captured = {}
    if enc == "whisper":
        feats = m.feature_extractor(wav, sampling_rate=16000, return_tensors="pt")["input_features"]
        captured["logmel"] = feats.numpy()
        hidden = m.encoder(feats).last_hidden_state
    else:
        feats = m.feature_extractor(wav, sampling_rate=16000, return_tensors="pt")["input_values"]
        hidden = m.encoder(feats).last_hidden_state
    captured["hidden"] = hidden.numpy()
    logits, offsets = m(torch.from_numpy(wav), torch.from_numpy(lang))
    probs = torch.softmax(logits, -1)
    maxp, arg = probs.max(-1)
    top2 = logits.topk(2, -1).values
    out = dict(
        lang_id=lang, seed=np.int64(seed), clip0=np.int64(clip0), L=np.int64(L), n_phonemes=np.int64(n_phonemes),
        argmax=arg.numpy().astype(np.int16), maxprob=maxp.numpy().astype(np.float32),
        margin=(top2[..., 0] - top2[..., 1]).numpy().astype(np.float32),
        offsets=offsets.numpy().astype(np.float32), bf16_weights=np.int64(int(bf16_weights)),
    )
    T = logits.shape[1]
    if full:
        out["logits"] = logits.numpy()
        out["hidden"] = captured["hidden"]
        if "logmel" in captured:
            out["logmel"] = captured["logmel"]
    else:
        ridx = np.unique(np.linspace(0, T - 1, rows).astype(np.int64))
        out["rows"] = ridx
        out["logits_rows"] = logits.numpy()[:, ridx]
        out["hidden_rows"] = captured["hidden"][:, ridx]
        out["hidden_abs_mean"] = np.abs(captured["hidden"]).mean(axis=(1, 2)).astype(np.float32)
        if "logmel" in captured:
            fidx = np.unique(np.linspace(0, captured["logmel"].shape[2] - 1, 40).astype(np.int64))
            out["logmel_frames"] = fidx
            out["logmel_cols"] = captured["logmel"][:, :, fidx]
            out["logmel_sum"] = captured["logmel"].astype(np.float64).sum(axis=(1, 2))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    thr = config["postprocess"]["confidence_threshold"]
    print(f"{name}: logits {tuple(logits.shape)} std {logits.std():.3f} | maxprob<{thr}: "
          f"{(maxp < thr).float().mean():.3f} | margin<0.05: {((top2[...,0]-top2[...,1]) < 0.05).float().mean():.4f}"
          f" | distinct argmax {len(np.unique(arg.numpy()))}")


def tiny_whisper_config(**kw):
    cfg = synth.base_config("whisper", whisper_model="local/whisper-tinytest", **kw)
    cfg["model"]["encoder_arch"] = dict(d_model=64, layers=2, heads=2, ffn=128, n_mels=80, max_positions=100)
    return cfg


def tiny_wavlm_config(stable, **kw):
    cfg = synth.base_config("wavlm", wavlm_model="local/wavlm-tinytest", **kw)
    cfg["model"]["encoder_arch"] = dict(
        d_model=64, layers=2, heads=2, ffn=128, conv_dim=(32,) * 7,
        feat_extract_norm="layer" if stable else "group", conv_bias=stable, stable_layer_norm=stable,
        pos_conv_kernel=16, pos_conv_groups=4, do_normalize=stable)
    return cfg


def postprocess_fixture():
    rng = np.random.RandomState(1234)
    phs = ["a", "b", "c", "SP", "AP"]
    labels = sorted(["O"] + [f"B-{p}" for p in phs] + [f"I-{p}" for p in phs])
    id2label = dict(enumerate(labels))
    cases = []
    for ci in range(12):
        T = int(rng.randint(5, 60))
        # tag sequences with realistic runs plus noise
        tags = []
        while len(tags) < T:
            r = rng.rand()
            if r < 0.25:
                tags += ["O"] * int(rng.randint(1, 4))
            else:
                p = phs[rng.randint(len(phs))]
                first = "B-" if rng.rand() < 0.8 else "I-"
                tags += [first + p] + ["I-" + p] * int(rng.randint(0, 5))
        tags = tags[:T]
        offs = rng.rand(T, 2).astype(np.float32)
        seg_no = ref_utils.decode_bio_tags(tags, frame_duration=0.02, offsets=None)
        seg_off = ref_utils.decode_bio_tags(tags, frame_duration=0.02, offsets=torch.from_numpy(offs))
        merged = {mode: ref_utils.merge_adjacent_segments(list(seg_off), mode=mode)
                  for mode in ("right", "left", "previous", "none")}
        forced = [s[2] for s in seg_off]
        if forced:
            forced = forced[: max(1, len(forced) - 1)] + ["zz"]
        aligned = ref_infer.align_phoneme_list(seg_off, forced) if forced else []
        with tempfile.NamedTemporaryFile("r", suffix=".lab", delete=False) as f:
            p = f.name
        ref_utils.save_lab(p, seg_off)
        lab = open(p, encoding="utf-8").read()
        os.unlink(p)
        ids = [labels.index(t) for t in tags]
        med = {str(k): [int(v) for v in median_filter(ids, size=k)] for k in (2, 3, 4, 5, 7)}
        logits = (rng.randn(T, len(labels)) * 2.0).astype(np.float32)
        sup = {str(th): ref_infer.suppress_low_confidence(torch.from_numpy(logits), id2label, threshold=th)
               for th in (0.0, 0.3, 0.5, 0.9)}
        cases.append(dict(tags=tags, offsets=offs.tolist(), segments_no_offsets=seg_no, segments=seg_off,
                          merged=merged, forced=forced, aligned=aligned, lab=lab, ids=ids, median=med,
                          logits=logits.tolist(), suppressed=sup))
    splits = {str(n): [len(s) for s in ref_infer.split_audio(np.zeros(n), 16000)]
              for n in (0, 1, 479999, 480000, 480001, 1040000)}
    mm = {"a": {"en": "ah", "ja": "a"}, "SP": {"en": "sil"}}
    c2l = [[ph, lang, ref_utils.canonical_to_lang(ph, lang, mm)]
           for ph in ("a", "SP", "q") for lang in ("en", "ja", "zz")]
    with open(os.path.join(HERE, "postprocess.json"), "w", encoding="utf-8") as f:
        json.dump(dict(labels=labels, cases=cases, split_audio=splits, merge_map=mm, canonical_to_lang=c2l), f)
    print("postprocess.json:", len(cases), "cases")


def metrics_fixture():
    """/root/reference/train.py:89-148 and correct_label.py:40-87, run as they are on seeded inputs -> metrics.json."""
    for name in ("pytorch_optimizer", "librosa", "torch.utils.tensorboard"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["torch.utils.tensorboard"].SummaryWriter = object
    import train as ref_train                      # noqa: E402  (the reference, unmodified)
    import correct_label as ref_cl                 # noqa: E402
    rng = np.random.RandomState(4321)
    names = ["a", "b", "en/c", "d", "ja/c", "e", "SP", "en/a"]

    def draw(nmax=14):
        n = int(rng.randint(0, nmax))
        t = np.sort(rng.uniform(0, 3, size=2 * n)).reshape(n, 2) if n else np.zeros((0, 2))
        return [[float(a), float(b), names[int(rng.randint(len(names)))]] for a, b in t]

    seg_cases = []
    for _ in range(240):
        p, g = draw(), draw()
        pt, gt = [tuple(x) for x in p], [tuple(x) for x in g]
        seg_cases.append(dict(pred=p, gt=g, per=float(ref_train.compute_phoneme_error_rate(pt, gt)),
                              ter=float(ref_train.compute_timing_error(pt, gt))))
    acc_cases = []
    for _ in range(40):
        B, T, C = int(rng.randint(1, 4)), int(rng.randint(0, 30)), int(rng.randint(2, 12))
        lg = rng.randn(B, T, C).astype(np.float32)
        lb = rng.randint(0, C, size=(B, T)).astype(np.int64)
        if T and rng.rand() < 0.5:                  # make some rows right
            lb[0] = lg[0].argmax(-1)
        acc_cases.append(dict(logits=lg.tolist(), labels=lb.tolist(),
                              acc=float(ref_train.compute_framewise_accuracy(torch.from_numpy(lg), torch.from_numpy(lb)))))
    clean = [[x, ref_train.clean_lab(x)] for x in ["en/AA", "k", "ja/zh/x", ""]]
    clean += [[[0.0, 1.0, "ja/k"], ref_train.clean_lab((0.0, 1.0, "ja/k"))], [[0.0, 1.0, ["en/t"]], ref_train.clean_lab((0.0, 1.0, ["en/t"]))]]
    snap_cases = []
    with tempfile.TemporaryDirectory() as tmp:
        for ci in range(220):
            n = int(rng.randint(0, 16))
            cuts = np.sort(rng.uniform(0, 4, size=n + 1))
            segs = [(float(cuts[i]), float(cuts[i + 1]), names[int(rng.randint(len(names)))].split("/")[-1]) for i in range(n)]
            if n and rng.rand() < 0.3:              # gaps between some segments
                segs = [(s + 0.004 * (i % 3), e, ph) for i, (s, e, ph) in enumerate(segs)]
            wav = os.path.join(tmp, "c%d.wav" % ci)
            lines = ["%d %d %s" % (int(s * 1e7), int(e * 1e7), ph) for s, e, ph in segs]
            if ci % 7 == 0:
                lines.insert(len(lines) // 2, "malformed line here too many")     # skipped by the reference (len(parts) != 3)
            if ci % 11 != 10:                        # (every 11th case has no .lab at all)
                with open(wav.replace(".wav", ".lab"), "w") as f:
                    f.write("\n".join(lines) + ("\n" if lines else ""))
            m = int(rng.randint(0, 24))
            pred = rng.uniform(0, 4, size=m)
            # half of the detections sit near real boundaries (inside and just outside the threshold), some are duplicates
            for k in range(m):
                if n and rng.rand() < 0.5:
                    pred[k] = cuts[int(rng.randint(n + 1))] + rng.choice([-0.031, -0.03, -0.012, 0.0, 0.007, 0.0299, 0.03, 0.045])
            pred = [float(x) for x in (np.sort(pred) if rng.rand() < 0.8 else pred)]
            if m > 2 and rng.rand() < 0.3:
                pred[1] = pred[0]
            thr = [0.03, 0.03, 0.01, 0.05][ci % 4]
            snapped, orig = ref_cl.correct_lab_boundaries(wav, list(pred), snap_threshold=thr)
            snap_cases.append(dict(lab_lines=lines if ci % 11 != 10 else None, predicted=pred, snap_threshold=thr,
                                   snapped=[list(x) for x in snapped], original=[list(x) for x in orig]))
    with open(os.path.join(HERE, "metrics.json"), "w", encoding="utf-8") as f:
        json.dump(dict(segments=seg_cases, accuracy=acc_cases, clean_lab=clean, snapping=snap_cases,
                       default_snap_threshold=ref_cl.snap_threshold_sec), f)
    print("metrics.json:", len(seg_cases), "segment cases,", len(acc_cases), "accuracy cases,", len(snap_cases), "snapping cases")


def main():
    which = set(sys.argv[1:])
    with tempfile.TemporaryDirectory() as tmp:
        if not which or "metrics" in which:
            metrics_fixture()
        if not which or "post" in which:
            postprocess_fixture()
        if not which or "tiny" in which:
            run_case("whisper_tiny", tiny_whisper_config(), 5, B=2, L=24000, seed=11, tmp=tmp, full=True)
        if not which or "cfg2" in which:
            run_case("whisper_base_cfg2", synth.baseline_config(1), 70, B=2, L=480000, seed=1, tmp=tmp, clip0=1000)
        if not which or "cfg2w" in which:
            # the same case on the bf16-rounded checkpoint: the target a bf16-WEIGHT deployment is held to with the tight tolerance
            run_case("whisper_base_cfg2_bf16w", synth.baseline_config(1), 70, B=2, L=480000, seed=1, tmp=tmp, clip0=1000,
                     bf16_weights=True)
        if not which or "full" in which:
            run_case("whisper_base_full", synth.base_config("whisper"), 70, B=1, L=300000, seed=2, tmp=tmp, clip0=2000)
        if not which or "wavlm" in which:
            run_case("wavlm_base_cfg1", synth.baseline_config(0), 4, B=1, L=16000, seed=3, tmp=tmp, full=True, sine=True)
            run_case("wavlm_tiny_group", tiny_wavlm_config(False), 5, B=2, L=12000, seed=4, tmp=tmp, full=True)
            run_case("wavlm_tiny_stable", tiny_wavlm_config(True), 5, B=2, L=12000, seed=5, tmp=tmp, full=True)


if __name__ == "__main__":
    main()

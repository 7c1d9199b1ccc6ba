"""Deterministic synthetic checkpoints and clips (no files ship, no torch RNG).  TEST / BENCH INFRASTRUCTURE, not product: it lives outside
the `wfl-asr_amd/` package (round 4; it used to be `wfl_asr_amd.synth`) and is imported by bench.py, tests/, tools/ and __graft_entry__.smoke().

There is no network here, so no pretrained/fine-tuned weights exist in this pipeline
(SURVEY.md §8c).  Benchmarks, golden fixtures and GPU parity tests therefore all run on a
*generated* checkpoint whose tensors follow the reference's state-dict naming and shapes
(/root/reference/model.py:54-146 + the HF encoder modules it wraps) and on generated 16 kHz
clips.  Everything is produced by an integer counter PRNG (splitmix64 over
`fnv1a(name) ^ seed + index`) so the GPU box regenerates bit-identical tensors.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np

import wfl_asr_amd  # noqa: F401  (registers the package from ./wfl-asr_amd)
from wfl_asr_amd.archs import WhisperArch, WavLMArch, head_config, resolve_encoder_arch  # noqa: F401 (head_config re-exported)

_M64 = (1 << 64) - 1


def _fnv1a64(s: str) -> int:
    h = 0xCBF29CE484222325
    for b in s.encode("utf-8"):
        h ^= b
        h = (h * 0x100000001B3) & _M64
    return h


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform01(name: str, n: int, seed: int = 0) -> np.ndarray:
    """n float64 values in [0,1), a pure function of (name, seed)."""
    base = np.uint64((_fnv1a64(name) ^ (seed * 0x9E3779B97F4A7C15)) & _M64)
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64) + base
    z = _splitmix64(_splitmix64(idx))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def _sym(name, shape, amp, seed):
    n = int(np.prod(shape))
    return ((uniform01(name, n, seed) * 2.0 - 1.0) * amp).astype(np.float32).reshape(shape)


# --------------------------------------------------------------------------------------
# state-dict specification (names/shapes exactly as the reference's strict load expects)
# --------------------------------------------------------------------------------------

def _lin(spec, prefix, out_f, in_f, bias=True, gain=1.0):
    spec[prefix + ".weight"] = ((out_f, in_f), "w", in_f, gain)
    if bias:
        spec[prefix + ".bias"] = ((out_f,), "b", in_f, 1.0)


def _ln(spec, prefix, d):
    spec[prefix + ".weight"] = ((d,), "ln_w", 0, 1.0)
    spec[prefix + ".bias"] = ((d,), "ln_b", 0, 1.0)


def _conv(spec, prefix, cout, cin_per_group, k, bias=True, gain=1.0):
    spec[prefix + ".weight"] = ((cout, cin_per_group, k), "w", cin_per_group * k, gain)
    if bias:
        spec[prefix + ".bias"] = ((cout,), "b", cin_per_group * k, 1.0)


def whisper_spec(a: WhisperArch) -> OrderedDict:
    s = OrderedDict()
    d = a.d_model
    _conv(s, "encoder.conv1", d, a.n_mels, 3)
    _conv(s, "encoder.conv2", d, d, 3)
    s["encoder.embed_positions.weight"] = ((a.max_positions, d), "pos", 0, 1.0)
    for i in range(a.layers):
        p = f"encoder.layers.{i}."
        _lin(s, p + "self_attn.k_proj", d, d, bias=False)
        _lin(s, p + "self_attn.v_proj", d, d)
        _lin(s, p + "self_attn.q_proj", d, d)
        _lin(s, p + "self_attn.out_proj", d, d)
        _ln(s, p + "self_attn_layer_norm", d)
        _lin(s, p + "fc1", a.ffn, d)
        _lin(s, p + "fc2", d, a.ffn)
        _ln(s, p + "final_layer_norm", d)
    _ln(s, "encoder.layer_norm", d)
    return s


def wavlm_spec(a: WavLMArch) -> OrderedDict:
    s = OrderedDict()
    d = a.d_model
    # no `masked_spec_embed`: the reference zeroes mask_time_prob (model.py:76-78), so HF never creates it
    cin = 1
    for i, (c, k) in enumerate(zip(a.conv_dim, a.conv_kernel)):
        p = f"encoder.feature_extractor.conv_layers.{i}."
        _conv(s, p + "conv", c, cin, k, bias=a.conv_bias, gain=1.6)
        if a.feat_extract_norm == "layer" or i == 0:
            _ln(s, p + "layer_norm", c)
        cin = c
    _ln(s, "encoder.feature_projection.layer_norm", a.conv_dim[-1])
    _lin(s, "encoder.feature_projection.projection", d, a.conv_dim[-1])
    pc = "encoder.encoder.pos_conv_embed.conv."
    s[pc + "bias"] = ((d,), "b", d // a.pos_conv_groups * a.pos_conv_kernel, 1.0)
    s[pc + "parametrizations.weight.original0"] = ((1, 1, a.pos_conv_kernel), "wn_g", 0, 1.0)
    s[pc + "parametrizations.weight.original1"] = (
        (d, d // a.pos_conv_groups, a.pos_conv_kernel), "w", d // a.pos_conv_groups * a.pos_conv_kernel, 1.0)
    _ln(s, "encoder.encoder.layer_norm", d)
    hd = d // a.heads
    for i in range(a.layers):
        p = f"encoder.encoder.layers.{i}."
        s[p + "attention.gru_rel_pos_const"] = ((1, a.heads, 1, 1), "ln_w", 0, 1.0)
        for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
            _lin(s, p + "attention." + n, d, d)
        _lin(s, p + "attention.gru_rel_pos_linear", 8, hd)
        if i == 0:
            s[p + "attention.rel_attn_embed.weight"] = ((a.num_buckets, a.heads), "relb", 0, 1.0)
        _ln(s, p + "layer_norm", d)
        _lin(s, p + "feed_forward.intermediate_dense", a.ffn, d)
        _lin(s, p + "feed_forward.output_dense", d, a.ffn)
        _ln(s, p + "final_layer_norm", d)
    return s


def head_spec(d: int, num_classes: int, h: dict, cls_gain: float = 6.0) -> OrderedDict:
    s = OrderedDict()
    e = h["lang_emb_dim"]
    s["lang_emb.weight"] = ((h["num_languages"], e), "emb", 0, 1.0)
    _lin(s, "lang_proj", d, d + e)
    if h["enable_bilstm"]:
        H = d // 2
        for layer in range(h["bilstm_num_layer"]):
            for suf in ("", "_reverse"):
                in_f = d if layer == 0 else 2 * H
                s[f"bilstm.weight_ih_l{layer}{suf}"] = ((4 * H, in_f), "lstm", H, 1.0)
                s[f"bilstm.weight_hh_l{layer}{suf}"] = ((4 * H, H), "lstm", H, 1.0)
                s[f"bilstm.bias_ih_l{layer}{suf}"] = ((4 * H,), "lstm", H, 1.0)
                s[f"bilstm.bias_hh_l{layer}{suf}"] = ((4 * H,), "lstm", H, 1.0)
    x = h["conformer_ff_expansion"]
    k = h["conformer_kernel_size"]
    for i in range(h["num_conformer_layers"]):
        p = f"conformer_layers.{i}."
        for ff in ("ff1", "ff2"):
            _ln(s, p + ff + ".net.0", d)
            _lin(s, p + ff + ".net.1", d * x, d)
            _lin(s, p + ff + ".net.4", d, d * x)
        s[p + "self_attn.in_proj_weight"] = ((3 * d, d), "w", d, 1.0)
        s[p + "self_attn.in_proj_bias"] = ((3 * d,), "b", d, 1.0)
        _lin(s, p + "self_attn.out_proj", d, d)
        _ln(s, p + "ln1", d)
        _ln(s, p + "ln2", d)
        _conv(s, p + "conv.0", 2 * d, d, 1)
        _conv(s, p + "conv.2", d, d, k)
        s[p + "conv.3.weight"] = ((d,), "ln_w", 0, 1.0)
        s[p + "conv.3.bias"] = ((d,), "ln_b", 0, 1.0)
        s[p + "conv.3.running_mean"] = ((d,), "ln_b", 0, 1.0)
        s[p + "conv.3.running_var"] = ((d,), "bn_var", 0, 1.0)
        s[p + "conv.3.num_batches_tracked"] = ((), "nbt", 0, 1.0)
        _conv(s, p + "conv.5", d, d, 1)
    if h["enable_dilated_conv"]:
        for i in range(h["dilated_conv_depth"]):
            _conv(s, f"dilated_conv_stack.{2 * i}", d, d, h["dilated_conv_kernel"], gain=1.4)
    # sharpened classifier (SURVEY.md §7 hard part 1): margins >> bf16 error, max-prob straddles 0.5
    _lin(s, "classifier", num_classes, d, gain=cls_gain)
    _conv(s, "boundary_offset_head.0", d, d, 3)
    _conv(s, "boundary_offset_head.2", 2, d, 1, gain=2.0)
    return s


def state_dict_spec(config: dict, num_classes: int, cls_gain: float = 6.0) -> OrderedDict:
    enc, arch = resolve_encoder_arch(config["model"], config.get("data"))
    s = whisper_spec(arch) if enc == "whisper" else (wavlm_spec(arch) if enc == "wavlm" else OrderedDict())
    s.update(head_spec(arch.d_model, num_classes, head_config(config["model"]), cls_gain))
    return s


def make_state_dict(config: dict, num_classes: int, seed: int = 0, cls_gain: float = 6.0) -> "OrderedDict[str, np.ndarray]":
    """name -> numpy array (float32; `num_batches_tracked` int64), reference naming."""
    out = OrderedDict()
    for name, (shape, kind, fan_in, gain) in state_dict_spec(config, num_classes, cls_gain).items():
        if kind == "w":
            t = _sym(name, shape, gain * (3.0 / fan_in) ** 0.5, seed)
        elif kind == "b":
            t = _sym(name, shape, 0.5 / max(fan_in, 1) ** 0.5, seed)
        elif kind == "lstm":
            t = _sym(name, shape, 1.0 / fan_in ** 0.5, seed)
        elif kind == "ln_w":
            t = 1.0 + _sym(name, shape, 0.1, seed)
        elif kind == "ln_b":
            t = _sym(name, shape, 0.05, seed)
        elif kind == "bn_var":
            t = 1.0 + _sym(name, shape, 0.5, seed)
        elif kind == "pos":
            t = _sym(name, shape, 0.5, seed)
        elif kind == "emb":
            t = _sym(name, shape, 1.0, seed)
        elif kind == "relb":
            t = _sym(name, shape, 1.0, seed)
        elif kind == "wn_g":
            t = (1.0 + _sym(name, shape, 0.2, seed)) * 1.5
        elif kind == "nbt":
            t = np.array(100, dtype=np.int64)
        else:
            raise AssertionError(kind)
        out[name] = t
    return out


def bf16_round(x: np.ndarray) -> np.ndarray:
    """float32 -> nearest bfloat16 (ties to even), returned as float32; NaN/inf pass through."""
    a = np.ascontiguousarray(x, dtype=np.float32)
    u = a.view(np.uint32)
    r = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)).view(np.float32)
    return np.where(np.isfinite(a), r, a).astype(np.float32)


_E4M3_POS = np.array([(m / 8.0) * 2.0 ** -6 if e == 0 else (1.0 + m / 8.0) * 2.0 ** (e - 7) for e in range(16) for m in range(8)][:127])


def fp8_round_rows(w: np.ndarray) -> np.ndarray:
    """[N, K] float32 -> the values an OCP e4m3 copy with one scale per row (row max -> 448) represents, as float32:
    what csrc/model.hip's pack8 stores (round to nearest, ties to even code)."""
    w = np.asarray(w, np.float32)
    mx = np.abs(w).max(axis=1, keepdims=True)
    scale = np.where(mx > 0, mx / np.float32(448.0), np.float32(1.0)).astype(np.float32)
    a = (np.abs(w) / scale).astype(np.float32)
    hi = np.clip(np.searchsorted(_E4M3_POS, a, side="right"), 1, 126)
    lo = hi - 1
    dl, dh = a - _E4M3_POS[lo], _E4M3_POS[hi] - a
    pick = np.where(dl < dh, lo, np.where(dh < dl, hi, np.where(lo % 2 == 1, hi, lo)))
    pick = np.where(a >= 448.0, 126, pick)
    return (np.sign(w) * _E4M3_POS[pick].astype(np.float32) * scale).astype(np.float32)


def round_weights_fp8(sd_np):
    """The "fp8 checkpoint" of a Whisper state dict: q / k / v / out_proj / fc1 / fc2 weights of every encoder layer replaced by
    their e4m3 + per-output-channel-scale values (model.weight_dtype: fp8), everything else as given."""
    out = OrderedDict()
    for k, v in sd_np.items():
        hit = k.startswith("encoder.layers.") and k.endswith(".weight") and any(
            t in k for t in ("q_proj", "k_proj", "v_proj", "out_proj", ".fc1.", ".fc2."))
        out[k] = fp8_round_rows(v) if hit else v
    return out


def round_weights_bf16(sd_np):
    """The "bf16 checkpoint" of a state dict: every float tensor with two or more dimensions (Linear / Conv / embedding
    weights -- what the MI355X path keeps in bf16) rounded to bfloat16 and stored back as float32; biases, norm parameters
    and BatchNorm statistics untouched.  The reference run on THIS checkpoint is the apples-to-apples target for a bf16-weight
    deployment: what is left between the two is activation rounding only (tests/golden/*_bf16w.npz)."""
    out = OrderedDict()
    for k, v in sd_np.items():
        out[k] = bf16_round(v) if (v.dtype == np.float32 and v.ndim >= 2) else v
    return out


# --------------------------------------------------------------------------------------
# synthetic clips
# --------------------------------------------------------------------------------------

def make_clip(index: int, n_samples: int, sr: int = 16000, seed: int = 0) -> np.ndarray:
    """Clip `index`: 3 gated sinusoids + 1 % uniform noise, peak-normalised the way
    /root/reference/infer.py:234-235 does (float64 `x / (max|x| + 1e-8)`), returned as float32."""
    tag = f"clip{index}"
    u = uniform01(tag + ".par", 6, seed)
    f = 80.0 + u[:3] * (4000.0 - 80.0)
    ph = u[3:] * 2.0 * np.pi
    t = np.arange(n_samples, dtype=np.float64) / sr
    # on/off gating with segment lengths 50-400 ms
    n_seg = int(n_samples / (0.05 * sr)) + 2
    seg_len = ((0.05 + 0.35 * uniform01(tag + ".seg", n_seg, seed)) * sr).astype(np.int64)
    edges = np.cumsum(seg_len)
    seg_of = np.searchsorted(edges, np.arange(n_samples), side="right")
    gate = (uniform01(tag + ".gate", n_seg + 1, seed) > 0.35).astype(np.float64)
    env = gate[seg_of]
    x = np.zeros(n_samples, dtype=np.float64)
    for k in range(3):
        # each partial gets its own segment-wise amplitude so segments differ spectrally
        ak = uniform01(tag + f".amp{k}", n_seg + 1, seed)[seg_of]
        x += ak * np.sin(2.0 * np.pi * f[k] * t + ph[k])
    x = 0.5 * x * env + 0.01 * (uniform01(tag + ".noise", n_samples, seed) * 2.0 - 1.0)
    if n_samples > 0:
        x = x / (np.max(np.abs(x)) + 1e-8)
    return x.astype(np.float32)


def make_batch(start: int, count: int, n_samples: int, sr: int = 16000, seed: int = 0) -> np.ndarray:
    return np.stack([make_clip(start + i, n_samples, sr, seed) for i in range(count)])


def sine_clip(n_samples: int = 16000, freq: float = 440.0, amp: float = 0.5, sr: int = 16000) -> np.ndarray:
    """BASELINE config 1's input: a pure sine."""
    t = np.arange(n_samples, dtype=np.float64) / sr
    return (amp * np.sin(2.0 * np.pi * freq * t)).astype(np.float32)


def make_labels(num_phonemes: int = 70) -> list:
    """A phonemes.txt-style label list: sorted, 2*P+1 entries (/root/reference/preprocess.py:165)."""
    phs = [f"p{i:02d}" for i in range(num_phonemes)]
    return sorted(["O"] + [f"B-{p}" for p in phs] + [f"I-{p}" for p in phs])


def base_config(encoder_type="whisper", whisper_model="openai/whisper-base",
                wavlm_model="microsoft/wavlm-base-plus", **model_overrides) -> dict:
    """A config dict with the reference's config.yaml keys (/root/reference/config.yaml:1-71)."""
    cfg = {
        "data": {"sample_rate": 16000, "frame_duration": 0.02, "n_mels": 80},
        "model": {
            "encoder_type": encoder_type,
            "whisper_model": whisper_model,
            "wavlm_model": wavlm_model,
            "freeze_encoder": False,
            "enable_bilstm": True,
            "bilstm_num_layer": 2,
            "enable_dilated_conv": True,
            "dilated_conv_depth": 2,
            "dilated_conv_kernel": 3,
            "num_conformer_layers": 2,
            "conformer_heads": 2,
            "conformer_ff_expansion": 2,
            "conformer_kernel_size": 31,
            "conformer_dropout": 0.15,
            "lang_emb_dim": 64,
            "num_languages": 2,
        },
        "output": {"save_dir": "."},
        "postprocess": {"median_filter": 1, "merge_segments": "right", "confidence_threshold": 0.5},
    }
    cfg["model"].update(model_overrides)
    return cfg


# the BASELINE.json configs, by index
def baseline_config(i: int) -> dict:
    if i == 0:   # cfg1: WavLM-base, linear head only
        return base_config("wavlm", wavlm_model="microsoft/wavlm-base", enable_bilstm=False,
                           num_conformer_layers=0, enable_dilated_conv=False)
    if i == 1:   # cfg2: Whisper-base + 2 Conformer
        return base_config("whisper", enable_bilstm=False, enable_dilated_conv=False)
    if i == 2:   # cfg3: WavLM-large + BiLSTM + dilated conv
        return base_config("wavlm", wavlm_model="microsoft/wavlm-large", num_conformer_layers=0)
    if i == 3:   # cfg4: Whisper-small + full head
        return base_config("whisper", whisper_model="openai/whisper-small")
    if i == 4:   # cfg5: Whisper-large-v3 encoder, fp8 weights (model.weight_dtype is this build's key: e4m3 + per-channel scale)
        return base_config("whisper", whisper_model="openai/whisper-large-v3", enable_bilstm=False,
                           num_conformer_layers=0, enable_dilated_conv=False, weight_dtype="fp8")
    raise IndexError(i)

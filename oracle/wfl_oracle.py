"""ORACLE — test infrastructure, not product code.

A CPU, fp32, pure-torch restatement of the WFL-ASR inference forward (no `transformers`
import, no reference import), used ONLY as the checker by `tests/`, `__graft_entry__.smoke()`
and `bench.py`'s `cpu_baseline` leg.  Nothing under `wfl-asr_amd/` may import this file.

Pinning: the reference has no tests or golden vectors of its own, so this
restatement is pinned against outputs of the reference itself, run in the build container by
`tests/golden/make_golden.py` (reference `BIOPhonemeTagger.forward` + HF transformers 5.15.0;
the reference pins 4.51.3 — version caveat in SURVEY.md §8c) and committed as fixtures under
`tests/golden/`; `tests/test_oracle_golden.py` checks every function here against them.

Exception: `mel_spectrogram_power` (the `encoder_type: none` front-end) restates torchaudio.transforms.MelSpectrogram from its
published definition -- torchaudio is not installed in this image and the reference holds no fixture for it, so that one function
is PARITY UNPINNED (the head behind it is the same pinned `head_forward`).

Every function cites the reference (or third-party) lines it follows:
  ref  = /root/reference/<file>
  HF   = transformers/<path> (5.15.0)
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F


# ---------------------------------------------------------------------------------------
# Whisper log-mel front-end       ref model.py:153-154 -> HF models/whisper/feature_extraction_whisper.py
# ---------------------------------------------------------------------------------------

def _hz_to_mel_slaney(f):
    # HF audio_utils.py:448-474 (mel_scale="slaney")
    f = np.asarray(f, dtype=np.float64)
    mels = 3.0 * f / 200.0
    logstep = 27.0 / np.log(6.4)
    reg = f >= 1000.0
    mels = np.where(reg, 15.0 + np.log(np.maximum(f, 1e-30) / 1000.0) * logstep, mels)
    return mels


def _mel_to_hz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    f = 200.0 * m / 3.0
    logstep = np.log(6.4) / 27.0
    reg = m >= 15.0
    return np.where(reg, 1000.0 * np.exp(logstep * (m - 15.0)), f)


def mel_filter_bank(n_mels: int, n_freq: int = 201, sr: int = 16000, fmax: float = 8000.0) -> np.ndarray:
    """[n_freq, n_mels] float64 Slaney-scale, Slaney-normalised triangles.
    HF audio_utils.py:638-729 with the arguments of feature_extraction_whisper.py:97-105;
    triangles per audio_utils.py:541-560."""
    mel_freqs = np.linspace(_hz_to_mel_slaney(0.0), _hz_to_mel_slaney(fmax), n_mels + 2)
    filter_freqs = _mel_to_hz_slaney(mel_freqs)
    fft_freqs = np.linspace(0, sr // 2, n_freq)
    diff = np.diff(filter_freqs)
    slopes = filter_freqs[None, :] - fft_freqs[:, None]
    down = -slopes[:, :-2] / diff[:-1]
    up = slopes[:, 2:] / diff[1:]
    fb = np.maximum(0.0, np.minimum(down, up))
    enorm = 2.0 / (filter_freqs[2:n_mels + 2] - filter_freqs[:n_mels])
    return fb * enorm[None, :]


def whisper_log_mel(wav: torch.Tensor, n_mels: int = 80, n_samples: int = 480000,
                    n_fft: int = 400, hop: int = 160) -> torch.Tensor:
    """[B, L] f32 -> [B, n_mels, n_samples // hop] f32.
    Pad/truncate to n_samples: HF feature_extraction_whisper.py:300-307;
    STFT/power/mel/log10/per-clip floor/affine: :135-168 (batched branch, per-clip max :159-161)."""
    wav = wav.to(torch.float32)
    B, L = wav.shape
    if L >= n_samples:
        wav = wav[:, :n_samples]
    else:
        wav = F.pad(wav, (0, n_samples - L))
    window = torch.hann_window(n_fft)
    stft = torch.stft(wav, n_fft, hop, window=window, return_complex=True)
    mag = (stft[..., :-1].abs() ** 2).contiguous()
    fb = torch.from_numpy(mel_filter_bank(n_mels, 1 + n_fft // 2)).to(torch.float32)
    mel = fb.T @ mag
    log_spec = torch.clamp(mel, min=1e-10).log10()
    mx = log_spec.amax(dim=(1, 2), keepdim=True)
    log_spec = torch.maximum(log_spec, mx - 8.0)
    return (log_spec + 4.0) / 4.0


# ---------------------------------------------------------------------------------------
# `encoder_type: none` front-end  ref model.py:82-91, 149-150 -> torchaudio.transforms.MelSpectrogram   (PARITY UNPINNED)
# ---------------------------------------------------------------------------------------

def mel_filter_bank_htk(n_mels: int, n_freq: int = 201, sr: int = 16000) -> np.ndarray:
    """torchaudio.functional.melscale_fbanks(n_freq, 0.0, sr / 2, n_mels, sr, norm=None, mel_scale="htk") -> [n_freq, n_mels] f32.
    m = 2595 log10(1 + f / 700); triangles of height 1 between consecutive mel-spaced points; computed in float32 like torchaudio."""
    all_freqs = torch.linspace(0, sr // 2, n_freq)
    m_min = 2595.0 * math.log10(1.0 + 0.0 / 700.0)
    m_max = 2595.0 * math.log10(1.0 + (sr / 2.0) / 700.0)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.clamp(torch.minimum(down, up), min=0.0).numpy()


def mel_spectrogram_power(wav: torch.Tensor, sr: int = 16000, n_fft: int = 400, hop: int = 320, n_mels: int = 80) -> torch.Tensor:
    """[B, L] f32 -> [B, n_mels, 1 + L // hop] f32: MelSpectrogram(sample_rate, n_fft, hop_length, n_mels) with torchaudio's defaults
    (win_length = n_fft, periodic Hann, center=True, pad_mode="reflect", power=2, normalized=False, f_min=0, f_max=sr/2, HTK mel
    scale, no filter normalisation, no log)."""
    wav = wav.to(torch.float32)
    spec = torch.stft(wav, n_fft, hop, n_fft, window=torch.hann_window(n_fft), center=True, pad_mode="reflect", normalized=False,
                      onesided=True, return_complex=True)
    power = spec.abs() ** 2
    fb = torch.from_numpy(mel_filter_bank_htk(n_mels, 1 + n_fft // 2, sr))
    return (power.transpose(-1, -2) @ fb).transpose(-1, -2)


# ---------------------------------------------------------------------------------------
# Whisper encoder                 ref model.py:155-156 -> HF models/whisper/modeling_whisper.py
# ---------------------------------------------------------------------------------------

def _ln(x, sd, p, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], eps)


def _linear(x, sd, p):
    return F.linear(x, sd[p + ".weight"], sd.get(p + ".bias"))


def _e4m3(x: torch.Tensor) -> torch.Tensor:
    """Round to OCP e4m3 (fn) and back: nearest even, |x| <= 448 expected."""
    return x.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(torch.float32)


def e4m3_rows(x: torch.Tensor) -> torch.Tensor:
    """x as the MI355X path's fp8 GEMMs see it: every row (last dim) scaled so that its maximum is 448, rounded to e4m3, scaled back
    (csrc/norm.hip, rows_fp8_kernel)."""
    mx = x.abs().amax(dim=-1, keepdim=True)
    scale = torch.where(mx > 0, mx * (1.0 / 448.0), torch.ones_like(mx))
    return _e4m3(x * (1.0 / scale)) * scale


def e4m3_blocks(x: torch.Tensor, block: int = 32) -> torch.Tensor:
    """x as an MX-fp8 operand (OCP microscaling: one e8m0 power-of-two scale per `block` consecutive elements of the last dim, elements
    e4m3): the block's scale is 2^(floor(log2(max|x|)) - 8), so the largest element lands in [256, 512) before the cast (448 is the
    format's largest finite value: the cast saturates the top of that octave, as the OCP spec's conversion does)."""
    sh = x.shape
    xb = x.reshape(*sh[:-1], sh[-1] // block, block)
    mx = xb.abs().amax(dim=-1, keepdim=True)
    e = torch.floor(torch.log2(torch.where(mx > 0, mx, torch.ones_like(mx)))) - 8.0
    e = e.clamp(-127.0, 127.0)
    scale = torch.exp2(e)
    return (_e4m3(xb / scale) * scale).reshape(sh)


def e4m3_pair_blocks(x: torch.Tensor, block: int = 32) -> torch.Tensor:
    """hi + lo: x rounded to block-scaled e4m3, plus the remainder rounded the same way (its own block scales) -- eight significant
    bits, what a bf16 operand carries, as two fp8 operands."""
    hi = e4m3_blocks(x, block)
    return hi + e4m3_blocks(x - hi, block)


ACT_FORMATS = {
    None: lambda t: t,
    "bf16": lambda t: t.to(torch.bfloat16).to(torch.float32),
    "row8": e4m3_rows,
    "fix8": lambda t: _e4m3(t * 8.0) * (1.0 / 8.0),
    "blk8": e4m3_blocks,
    "pair8": e4m3_pair_blocks,
}
# the four GEMM inputs of an encoder layer: LayerNorm 1 -> q|k|v, attention context -> out_proj, LayerNorm 2 -> fc1, GELU -> fc2
ACT_FP8_ROUND3 = {"ln1": "row8", "ctx": "fix8", "ln2": "row8", "gelu": "fix8"}


def whisper_encoder(feats: torch.Tensor, sd: dict, heads: int, layers: int, prefix: str = "encoder.", act_fp8=False) -> torch.Tensor:
    """[B, n_mels, 2T] -> [B, T, d].  HF modeling_whisper.py:618-642 (stem 618-625, layers 627-640,
    final LN 642); layer 391-407; attention 309 (q scaled), 332-333 (k no bias), SDPA scaling 1.0.
    act_fp8 (NOT the reference -- a diagnostic of what an activation format costs inside the reference's arithmetic; the parity target
    of the fp8 build is this function with act_fp8 = False on the fp8-rounded checkpoint): a dict naming the format (ACT_FORMATS) of
    each of the four GEMM inputs of a layer {"ln1", "ctx", "ln2", "gelu"}; True = round 3's choice (ACT_FP8_ROUND3; head size 64)."""
    if act_fp8 is True:
        act_fp8 = ACT_FP8_ROUND3
    fmt = {k: ACT_FORMATS[(act_fp8 or {}).get(k)] for k in ("ln1", "ctx", "ln2", "gelu")}
    p = prefix
    x = F.gelu(F.conv1d(feats, sd[p + "conv1.weight"], sd[p + "conv1.bias"], padding=1))
    x = F.gelu(F.conv1d(x, sd[p + "conv2.weight"], sd[p + "conv2.bias"], stride=2, padding=1))
    x = x.permute(0, 2, 1) + sd[p + "embed_positions.weight"]
    B, T, d = x.shape
    hd = d // heads
    for i in range(layers):
        lp = f"{p}layers.{i}."
        h = fmt["ln1"](_ln(x, sd, lp + "self_attn_layer_norm"))
        q = (_linear(h, sd, lp + "self_attn.q_proj") * hd ** -0.5).view(B, T, heads, hd).transpose(1, 2)
        k = _linear(h, sd, lp + "self_attn.k_proj").view(B, T, heads, hd).transpose(1, 2)
        v = _linear(h, sd, lp + "self_attn.v_proj").view(B, T, heads, hd).transpose(1, 2)
        a = torch.softmax(q @ k.transpose(2, 3), dim=-1) @ v
        a = fmt["ctx"](a.transpose(1, 2).reshape(B, T, d))
        x = x + _linear(a, sd, lp + "self_attn.out_proj")
        h = fmt["ln2"](_ln(x, sd, lp + "final_layer_norm"))
        h = fmt["gelu"](F.gelu(_linear(h, sd, lp + "fc1")))
        x = x + _linear(h, sd, lp + "fc2")
    return _ln(x, sd, p + "layer_norm")


# ---------------------------------------------------------------------------------------
# WavLM front-end + encoder        ref model.py:159-161 -> HF wav2vec2 feature extractor + modeling_wavlm.py
# ---------------------------------------------------------------------------------------

def wavlm_normalize(wav: torch.Tensor, do_normalize: bool) -> torch.Tensor:
    """HF models/wav2vec2/feature_extraction_wav2vec2.py:78-97 (no attention mask: whole clip)."""
    if not do_normalize:
        return wav
    m = wav.mean(dim=1, keepdim=True)
    v = wav.var(dim=1, unbiased=False, keepdim=True)
    return (wav - m) / torch.sqrt(v + 1e-7)


def wavlm_feature_encoder(wav: torch.Tensor, sd: dict, arch, prefix="encoder.") -> torch.Tensor:
    """[B, L] -> [B, T, C].  HF modeling_wavlm.py:675-744 (layer kinds), 772-782 (stack)."""
    x = wav[:, None, :]
    for i, (k, s) in enumerate(zip(arch.conv_kernel, arch.conv_stride)):
        lp = f"{prefix}feature_extractor.conv_layers.{i}."
        x = F.conv1d(x, sd[lp + "conv.weight"], sd.get(lp + "conv.bias"), stride=s)
        if arch.feat_extract_norm == "group":
            if i == 0:
                C = x.shape[1]
                x = F.group_norm(x, C, sd[lp + "layer_norm.weight"], sd[lp + "layer_norm.bias"], 1e-5)
        else:
            x = x.transpose(1, 2)
            x = F.layer_norm(x, (x.shape[-1],), sd[lp + "layer_norm.weight"], sd[lp + "layer_norm.bias"], 1e-5)
            x = x.transpose(1, 2)
        x = F.gelu(x)
    return x.transpose(1, 2)


def _wavlm_rel_buckets(T: int, num_buckets: int, max_distance: int) -> torch.Tensor:
    """HF modeling_wavlm.py:243-271 (bidirectional bucketing). -> [T, T] int64."""
    ctx = torch.arange(T)[:, None]
    mem = torch.arange(T)[None, :]
    rel = mem - ctx
    nb = num_buckets // 2
    buckets = (rel > 0).to(torch.long) * nb
    rel = rel.abs()
    max_exact = nb // 2
    is_small = rel < max_exact
    large = torch.log(rel.float() / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)
    large = (max_exact + large).to(torch.long)
    large = torch.min(large, torch.full_like(large, nb - 1))
    return buckets + torch.where(is_small, rel, large)


def wavlm_encoder(wav: torch.Tensor, sd: dict, arch, prefix="encoder.") -> torch.Tensor:
    """[B, L] (already feature-extractor-normalised) -> [B, T, d].
    HF modeling_wavlm.py:1032-1088 (model), 93-105 (projection), 48-90 (pos conv, weight-norm dim=2,
    drop last step), 147-271 (gated rel-pos attention), 388-447 / 465-522 (post-LN / stable pre-LN)."""
    p = prefix
    eps = arch.layer_norm_eps
    feats = wavlm_feature_encoder(wav, sd, arch, p)
    x = F.layer_norm(feats, (feats.shape[-1],), sd[p + "feature_projection.layer_norm.weight"],
                     sd[p + "feature_projection.layer_norm.bias"], eps)
    x = _linear(x, sd, p + "feature_projection.projection")
    B, T, d = x.shape
    # positional conv: weight = g * v / ||v||  with the norm over dims (0, 1) (weight_norm dim=2)
    pc = p + "encoder.pos_conv_embed.conv."
    g = sd[pc + "parametrizations.weight.original0"]
    v = sd[pc + "parametrizations.weight.original1"]
    w = g * v / v.norm(p=2, dim=(0, 1), keepdim=True)
    K = arch.pos_conv_kernel
    pos = F.conv1d(x.transpose(1, 2), w, sd[pc + "bias"], padding=K // 2, groups=arch.pos_conv_groups)
    if K % 2 == 0:
        pos = pos[:, :, :-1]
    x = x + F.gelu(pos).transpose(1, 2)
    if not arch.stable_layer_norm:
        x = F.layer_norm(x, (d,), sd[p + "encoder.layer_norm.weight"], sd[p + "encoder.layer_norm.bias"], eps)
    heads = arch.heads
    hd = d // heads
    buckets = _wavlm_rel_buckets(T, arch.num_buckets, arch.max_distance)
    rel_emb = sd[p + "encoder.layers.0.attention.rel_attn_embed.weight"]      # [num_buckets, heads]
    pos_bias = rel_emb[buckets].permute(2, 0, 1)                                # [heads, T, T]
    for i in range(arch.layers):
        lp = f"{p}encoder.layers.{i}."
        res = x
        h = _ln(x, sd, lp + "layer_norm", eps) if arch.stable_layer_norm else x
        # gated relative position bias: modeling_wavlm.py:167-180
        gh = h.view(B, T, heads, hd).permute(0, 2, 1, 3)
        proj = _linear(gh, sd, lp + "attention.gru_rel_pos_linear").view(B, heads, T, 2, 4).sum(-1)
        ga, gb = torch.sigmoid(proj).chunk(2, dim=-1)
        gate = ga * (gb * sd[lp + "attention.gru_rel_pos_const"] - 1.0) + 2.0   # [B, heads, T, 1]
        bias = gate * pos_bias[None]                                           # [B, heads, T, T]
        q = _linear(h, sd, lp + "attention.q_proj").view(B, T, heads, hd).transpose(1, 2) * hd ** -0.5
        k = _linear(h, sd, lp + "attention.k_proj").view(B, T, heads, hd).transpose(1, 2)
        vv = _linear(h, sd, lp + "attention.v_proj").view(B, T, heads, hd).transpose(1, 2)
        a = torch.softmax(q @ k.transpose(2, 3) + bias, dim=-1) @ vv
        a = _linear(a.transpose(1, 2).reshape(B, T, d), sd, lp + "attention.out_proj")
        x = res + a
        if arch.stable_layer_norm:
            h = _ln(x, sd, lp + "final_layer_norm", eps)
            h = _linear(F.gelu(_linear(h, sd, lp + "feed_forward.intermediate_dense")), sd, lp + "feed_forward.output_dense")
            x = x + h
        else:
            x = _ln(x, sd, lp + "layer_norm", eps)
            h = _linear(F.gelu(_linear(x, sd, lp + "feed_forward.intermediate_dense")), sd, lp + "feed_forward.output_dense")
            x = _ln(x + h, sd, lp + "final_layer_norm", eps)
    if arch.stable_layer_norm:
        x = F.layer_norm(x, (d,), sd[p + "encoder.layer_norm.weight"], sd[p + "encoder.layer_norm.bias"], eps)
    return x


def wavlm_num_frames(L: int, arch) -> int:
    """HF modeling_wavlm.py `_get_feat_extract_output_lengths`: floor((L - k) / s) + 1 per layer."""
    for k, s in zip(arch.conv_kernel, arch.conv_stride):
        L = (L - k) // s + 1
    return L


# ---------------------------------------------------------------------------------------
# Head                              ref model.py:176-194
# ---------------------------------------------------------------------------------------

def bilstm(x: torch.Tensor, sd: dict, num_layers: int) -> torch.Tensor:
    """nn.LSTM(bidirectional, batch_first) written out: ref model.py:104-111, 182-183.
    Gate order i, f, g, o; both biases added; zero initial state; concat(fwd, bwd)."""
    B, T, _ = x.shape
    for layer in range(num_layers):
        outs = []
        for suf, rev in (("", False), ("_reverse", True)):
            w_ih = sd[f"bilstm.weight_ih_l{layer}{suf}"]
            w_hh = sd[f"bilstm.weight_hh_l{layer}{suf}"]
            b = sd[f"bilstm.bias_ih_l{layer}{suf}"] + sd[f"bilstm.bias_hh_l{layer}{suf}"]
            H = w_hh.shape[1]
            gx = x @ w_ih.T + b                         # [B, T, 4H]
            h = x.new_zeros(B, H)
            c = x.new_zeros(B, H)
            ys = [None] * T
            order = range(T - 1, -1, -1) if rev else range(T)
            for t in order:
                g = gx[:, t] + h @ w_hh.T
                i_, f_, g_, o_ = g.chunk(4, dim=1)
                c = torch.sigmoid(f_) * c + torch.sigmoid(i_) * torch.tanh(g_)
                h = torch.sigmoid(o_) * torch.tanh(c)
                ys[t] = h
            outs.append(torch.stack(ys, dim=1))
        x = torch.cat(outs, dim=-1)
    return x


def _mha(x, sd, p, heads):
    """nn.MultiheadAttention(batch_first) self-attention, eval: packed in_proj, q scaled by hd^-1/2."""
    B, T, d = x.shape
    hd = d // heads
    qkv = F.linear(x, sd[p + ".in_proj_weight"], sd[p + ".in_proj_bias"])
    q, k, v = qkv.chunk(3, dim=-1)
    q = q.view(B, T, heads, hd).transpose(1, 2) * hd ** -0.5
    k = k.view(B, T, heads, hd).transpose(1, 2)
    v = v.view(B, T, heads, hd).transpose(1, 2)
    a = torch.softmax(q @ k.transpose(2, 3), dim=-1) @ v
    return _linear(a.transpose(1, 2).reshape(B, T, d), sd, p + ".out_proj")


def _ff(x, sd, p):
    # ref model.py:9-19: LN -> Linear -> GELU -> (Dropout) -> Linear -> (Dropout)
    h = _ln(x, sd, p + ".net.0")
    return _linear(F.gelu(_linear(h, sd, p + ".net.1")), sd, p + ".net.4")


def conformer_block(x: torch.Tensor, sd: dict, p: str, heads: int, kernel: int) -> torch.Tensor:
    """ref model.py:40-52 (modules 21-38).  BatchNorm1d in eval mode uses running stats."""
    x = x + 0.5 * _ff(x, sd, p + "ff1")
    x = _ln(x + _mha(x, sd, p + "self_attn", heads), sd, p + "ln1")
    h = _ln(x, sd, p + "ln2").transpose(1, 2)
    h = F.conv1d(h, sd[p + "conv.0.weight"], sd[p + "conv.0.bias"])
    h = F.glu(h, dim=1)
    h = F.conv1d(h, sd[p + "conv.2.weight"], sd[p + "conv.2.bias"], padding=kernel // 2)
    h = F.batch_norm(h, sd[p + "conv.3.running_mean"], sd[p + "conv.3.running_var"],
                     sd[p + "conv.3.weight"], sd[p + "conv.3.bias"], False, 0.0, 1e-5)
    h = F.gelu(h)
    h = F.conv1d(h, sd[p + "conv.5.weight"], sd[p + "conv.5.bias"]).transpose(1, 2)
    if x.size(1) != h.size(1):                      # model.py:46-49 (only for even kernels)
        m = min(x.size(1), h.size(1))
        x, h = x[:, :m], h[:, :m]
    x = x + h
    return x + 0.5 * _ff(x, sd, p + "ff2")


def head_forward(hidden: torch.Tensor, lang_id, sd: dict, hc: dict):
    """Encoder output [B, T, d] -> (logits [B, T, C], offsets [B, T, 2]).  ref model.py:176-194."""
    x = hidden
    if lang_id is not None:
        e = sd["lang_emb.weight"][lang_id]                            # [B, 64]
        e = e[:, None, :].expand(-1, x.size(1), -1)
        x = _linear(torch.cat([x, e], dim=-1), sd, "lang_proj")
    if hc["enable_bilstm"]:
        x = bilstm(x, sd, hc["bilstm_num_layer"])
    for i in range(hc["num_conformer_layers"]):
        x = conformer_block(x, sd, f"conformer_layers.{i}.", hc["conformer_heads"], hc["conformer_kernel_size"])
    if hc["enable_dilated_conv"]:
        h = x.transpose(1, 2)
        k = hc["dilated_conv_kernel"]
        for i in range(hc["dilated_conv_depth"]):
            dil = 2 ** i
            h = F.relu(F.conv1d(h, sd[f"dilated_conv_stack.{2 * i}.weight"], sd[f"dilated_conv_stack.{2 * i}.bias"],
                                dilation=dil, padding=dil * (k - 1) // 2))
        x = h.transpose(1, 2)
    logits = _linear(x, sd, "classifier")
    h = x.transpose(1, 2)
    h = F.gelu(F.conv1d(h, sd["boundary_offset_head.0.weight"], sd["boundary_offset_head.0.bias"], padding=1))
    h = torch.sigmoid(F.conv1d(h, sd["boundary_offset_head.2.weight"], sd["boundary_offset_head.2.bias"]))
    return logits, h.transpose(1, 2)


# ---------------------------------------------------------------------------------------
# Whole forward + tag decision
# ---------------------------------------------------------------------------------------

def to_torch_state_dict(sd_np: dict) -> dict:
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd_np.items()}


@torch.no_grad()
def forward(wav: torch.Tensor, lang_id, sd: dict, enc: str, arch, hc: dict, return_hidden: bool = False, act_fp8: bool = False):
    """ref model.py:148-194 BIOPhonemeTagger.forward (encoder_type whisper | wavlm | none).  act_fp8: see whisper_encoder."""
    if enc == "whisper":
        feats = whisper_log_mel(wav, arch.n_mels, arch.max_positions * 2 * arch.hop, arch.n_fft, arch.hop)
        hidden = whisper_encoder(feats, sd, arch.heads, arch.layers, act_fp8=act_fp8)
    elif enc == "wavlm":
        hidden = wavlm_encoder(wavlm_normalize(wav.to(torch.float32), arch.do_normalize), sd, arch)
    elif enc == "none":
        hidden = mel_spectrogram_power(wav, arch.sample_rate, arch.n_fft, arch.hop, arch.n_mels).transpose(1, 2)
    else:
        raise ValueError(enc)
    logits, offsets = head_forward(hidden, lang_id, sd, hc)
    if return_hidden:
        return logits, offsets, hidden
    return logits, offsets


@torch.no_grad()
def tags_from_logits(logits: torch.Tensor, o_id: int, threshold: float):
    """ref infer.py:86-96 + 169/297: softmax -> (max prob, argmax); prob < threshold => "O".
    Returns (ids int64 [.., T], maxprob f32, argmax int64, top-2 logit margin f32)."""
    probs = torch.softmax(logits, dim=-1)
    maxp, arg = probs.max(dim=-1)
    ids = torch.where(maxp < threshold, torch.full_like(arg, o_id), arg)
    top2 = logits.topk(2, dim=-1).values
    return ids, maxp, arg, top2[..., 0] - top2[..., 1]

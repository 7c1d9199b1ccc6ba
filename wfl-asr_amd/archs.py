"""Built-in encoder dimension table.

The reference gets these by fetching the HuggingFace config *by model name* every time a
model is built (/root/reference/model.py:69-70, 74-80).  Head count and mel-bin count are not
recoverable from a checkpoint (q_proj is [d, d] for any head count; the feature extractor is
not an nn.Module), so the build keeps the public values here, keyed by the same names the
reference's config.yaml uses (`model.whisper_model` / `model.wavlm_model`).  A config may
override any field through `model.encoder_arch: {...}`.
"""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass


@dataclass
class WhisperArch:
    d_model: int
    layers: int
    heads: int
    ffn: int
    n_mels: int = 80
    max_positions: int = 1500   # encoder output frames; mel frames = 2 * max_positions
    n_fft: int = 400
    hop: int = 160


@dataclass
class WavLMArch:
    d_model: int
    layers: int
    heads: int
    ffn: int
    conv_dim: tuple = (512,) * 7
    conv_kernel: tuple = (10, 3, 3, 3, 3, 2, 2)
    conv_stride: tuple = (5, 2, 2, 2, 2, 2, 2)
    feat_extract_norm: str = "group"      # "group" (base) | "layer" (large)
    conv_bias: bool = False
    stable_layer_norm: bool = False        # large: True
    pos_conv_kernel: int = 128
    pos_conv_groups: int = 16
    num_buckets: int = 320
    max_distance: int = 800
    do_normalize: bool = False             # from the hub preprocessor_config.json (base: False, large: True)
    layer_norm_eps: float = 1e-5


@dataclass
class MelArch:
    """`encoder_type: none` (/root/reference/model.py:82-91): the hidden states are a torchaudio MelSpectrogram, width n_mels."""
    d_model: int                 # = n_mels (model.py:91)
    n_mels: int = 80
    hop: int = 320               # int(frame_duration * sample_rate) (model.py:88)
    n_fft: int = 400
    sample_rate: int = 16000
    layers: int = 0
    heads: int = 1
    ffn: int = 0


WHISPER = {
    "tiny": WhisperArch(384, 4, 6, 1536),
    "base": WhisperArch(512, 6, 8, 2048),
    "small": WhisperArch(768, 12, 12, 3072),
    "medium": WhisperArch(1024, 24, 16, 4096),
    "large": WhisperArch(1280, 32, 20, 5120),
    "large-v2": WhisperArch(1280, 32, 20, 5120),
    "large-v3": WhisperArch(1280, 32, 20, 5120, n_mels=128),
    "large-v3-turbo": WhisperArch(1280, 32, 20, 5120, n_mels=128),
}

WAVLM = {
    "base": WavLMArch(768, 12, 12, 3072),
    "base-plus": WavLMArch(768, 12, 12, 3072),
    "base-sv": WavLMArch(768, 12, 12, 3072),
    "large": WavLMArch(1024, 24, 16, 4096, feat_extract_norm="layer", conv_bias=True,
                       stable_layer_norm=True, do_normalize=True),
}


def _suffix(name: str, prefix: str) -> str:
    base = name.rstrip("/").split("/")[-1].lower()
    if base.endswith(".en"):
        base = base[:-3]
    if base.startswith(prefix):
        base = base[len(prefix):]
    return base.lstrip("-_")


def resolve_encoder_arch(model_cfg: dict, data_cfg: dict | None = None):
    """config["model"] (+ config["data"]) -> ("whisper", WhisperArch) | ("wavlm", WavLMArch) | ("none", MelArch).

    Mirrors the selection at /root/reference/model.py:57-93.  `encoder_type: none` reads sample_rate, frame_duration and n_mels
    from config["data"] (model.py:85-90).
    """
    enc = str(model_cfg["encoder_type"]).lower()
    override = dict(model_cfg.get("encoder_arch") or {})
    if enc == "whisper":
        key = _suffix(str(model_cfg["whisper_model"]), "whisper")
        table, cls = WHISPER, WhisperArch
    elif enc == "wavlm":
        key = _suffix(str(model_cfg["wavlm_model"]), "wavlm")
        table, cls = WAVLM, WavLMArch
    elif enc in ("none", "null"):
        if data_cfg is None:
            raise ValueError("encoder_type 'none' needs config['data'] (sample_rate, frame_duration, n_mels)")
        sr = int(data_cfg["sample_rate"])
        if sr != 16000:
            raise ValueError("encoder_type 'none': the mel front-end is built for 16 kHz input (config.data.sample_rate)")
        n_mels = int(data_cfg.get("n_mels", 80))
        arch = MelArch(n_mels, n_mels, int(data_cfg.get("frame_duration", 0.02) * sr), 400, sr)
        if override:
            raise ValueError("model.encoder_arch does not apply to encoder_type 'none'")
        return "none", arch
    else:
        raise ValueError("Unsupported encoder type. Use 'whisper', 'wavlm', or 'none'.")
    if key in table:
        arch = dataclasses.replace(table[key])
    elif override and all(k in override for k in ("d_model", "layers", "heads", "ffn")):
        arch = cls(override["d_model"], override["layers"], override["heads"], override["ffn"])
    else:
        raise ValueError(
            f"unknown {enc} model '{key}': add model.encoder_arch {{d_model, layers, heads, ffn, ...}} to the config"
        )
    for k, v in override.items():
        if not hasattr(arch, k):
            raise ValueError(f"model.encoder_arch has no field '{k}'")
        if isinstance(getattr(arch, k), tuple):
            v = tuple(v)
        setattr(arch, k, v)
    return enc, arch


def head_config(model_cfg: dict) -> dict:
    """The `.get()` defaults scattered through /root/reference/model.py:61-66, 96, 108, 118-123."""
    return dict(
        enable_bilstm=bool(model_cfg.get("enable_bilstm", True)),
        bilstm_num_layer=int(model_cfg.get("bilstm_num_layer", 1)),
        enable_dilated_conv=bool(model_cfg.get("enable_dilated_conv", True)),
        dilated_conv_depth=int(model_cfg.get("dilated_conv_depth", 2)),
        dilated_conv_kernel=int(model_cfg.get("dilated_conv_kernel", 3)),
        num_conformer_layers=int(model_cfg.get("num_conformer_layers", 2)),
        conformer_heads=int(model_cfg.get("conformer_heads", 4)),
        conformer_ff_expansion=int(model_cfg.get("conformer_ff_expansion", 4)),
        conformer_kernel_size=int(model_cfg.get("conformer_kernel_size", 31)),
        lang_emb_dim=int(model_cfg.get("lang_emb_dim", 64)),
        num_languages=int(model_cfg["num_languages"]),
    )

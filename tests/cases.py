"""Configs of the golden cases (must match tests/golden/make_golden.py)."""
import synthetic as synth


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


GOLDEN_CASES = {
    "whisper_tiny": tiny_whisper_config,
    "whisper_base_cfg2": lambda: synth.baseline_config(1),
    "whisper_base_cfg2_bf16w": lambda: synth.baseline_config(1),     # reference run on the bf16-rounded checkpoint
    "whisper_base_full": lambda: synth.base_config("whisper"),
    "wavlm_base_cfg1": lambda: synth.baseline_config(0),
    "wavlm_tiny_group": lambda: tiny_wavlm_config(False),
    "wavlm_tiny_stable": lambda: tiny_wavlm_config(True),
}

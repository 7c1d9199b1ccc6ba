"""ctypes binding of libwfl_asr_hip.so (include/wfl_asr.h).  Fails loudly when the HIP library is missing:
there is no CPU or eager-PyTorch fallback for the model forward anywhere in this package."""
from __future__ import annotations

import ctypes as C
import os

PKG = os.path.dirname(os.path.abspath(__file__))
# WFL_LIB_PATH: an alternative build of the same library (A/B runs of compiler flags: tools/build_variant.sh); never a fallback
LIB_PATH = os.environ.get("WFL_LIB_PATH") or os.path.join(PKG, "libwfl_asr_hip.so")
ABI_VERSION = 2


class WflArch(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32), ("encoder_type", C.c_int32), ("d_model", C.c_int32), ("enc_layers", C.c_int32),
        ("enc_heads", C.c_int32), ("enc_ffn", C.c_int32), ("n_mels", C.c_int32), ("max_positions", C.c_int32),
        ("num_classes", C.c_int32), ("o_id", C.c_int32), ("num_languages", C.c_int32), ("lang_emb_dim", C.c_int32),
        ("enable_bilstm", C.c_int32), ("bilstm_layers", C.c_int32), ("n_conformer", C.c_int32),
        ("conformer_heads", C.c_int32), ("conformer_ff_expansion", C.c_int32), ("conformer_kernel", C.c_int32),
        ("enable_dilated", C.c_int32), ("dilated_depth", C.c_int32), ("dilated_kernel", C.c_int32),
        ("wavlm_n_conv", C.c_int32), ("wavlm_conv_dim", C.c_int32 * 8), ("wavlm_conv_kernel", C.c_int32 * 8),
        ("wavlm_conv_stride", C.c_int32 * 8), ("wavlm_group_norm", C.c_int32), ("wavlm_conv_bias", C.c_int32),
        ("wavlm_stable_layer_norm", C.c_int32), ("wavlm_pos_conv_kernel", C.c_int32),
        ("wavlm_pos_conv_groups", C.c_int32), ("wavlm_num_buckets", C.c_int32), ("wavlm_max_distance", C.c_int32),
        ("wavlm_do_normalize", C.c_int32), ("fp8_weights", C.c_int32), ("mel_hop", C.c_int32), ("precision", C.c_int32), ("fp8_activations", C.c_int32), ("reserved", C.c_int32 * 6),
    ]


_P = C.c_void_p
_I = C.c_int32
_L = C.c_int64
_F = C.c_float

# name -> (restype, argtypes); exactly the declarations of include/wfl_asr.h
SIGNATURES = {
    "wfl_last_error": (C.c_char_p, []),
    "wfl_abi_version": (_I, []),
    "wfl_create": (_I, [C.POINTER(WflArch), C.POINTER(_P)]),
    "wfl_destroy": (None, [_P]),
    "wfl_load_tensor": (_I, [_P, C.c_char_p, _P, C.POINTER(_L), _I]),
    "wfl_finalize": (_I, [_P]),
    "wfl_num_frames": (_I, [_P, _I]),
    "wfl_workspace_bytes": (_L, [_P, _I, _I]),
    "wfl_device": (_I, [_P]),
    "wfl_set_average_languages": (_I, [_P, _P, _I]),
    "wfl_forward": (_I, [_P, _P, _L, _P, _I, _I, _P, _I, _F, _P, _L, _P, _P, _P, _P, _P, _P, _P, _P]),
    "wfl_encode": (_I, [_P, _P, _L, _P, _I, _I, _P, _L, _P, _P]),
    "wfl_head_workspace_bytes": (_L, [_P, _I, _I]),
    "wfl_head": (_I, [_P, _P, _I, _I, _P, _I, _F, _P, _L, _P, _P, _P, _P, _P, _P, _P]),
    "wfl_check": (_I, [_P, _P, _L, _I, _I, _P]),
    "wfl_logmel": (_I, [_P, _P, _L, _P, _I, _I, _P, _P, _L, _P]),
    "wfl_op_gemm": (_I, [_P, _L, _I, _L, _P, _I, _I, _I, _I, _I, _I, _P, _L, _L, _I, _P, _P, _L, _F, _I, _I, _I, _P]),
    "wfl_op_gemm_split": (_I, [_P, _P, _L, _I, _L, _P, _I, _I, _I, _I, _I, _I, _P, _P, _L, _L, _I, _P, _P, _P, _L, _F, _I, _I, _P]),
    "wfl_op_gemm_ln": (_I, [_P, _L, _P, _I, _I, _I, _I, _I, _I, _P, _L, _L, _I, _P, _P, _F, _I, _P]),
    "wfl_op_gemm_mx": (_I, [_P, _P, _L, _P, _P, _P, _F, _I, _I, _I, _I, _I, _P, _L, _L, _I, _P, _P, _P, _P, _F, _I, _P, _P, _L, _F, _P, _P]),
    "wfl_op_rows_fp8": (_I, [_P, _L, _P, _P, _P, _F, _L, _I, _I, _I, _I, _P, _P, _L, _P, _P]),
    "wfl_op_attention": (_I, [_P, _L, _L, _P, _L, _P, _L, _I, _I, _I, _I, _I, _P]),
    "wfl_op_layernorm": (_I, [_P, _L, _P, _L, _P, _P, _F, _L, _I, _I, _I, _I, _P]),
    "wfl_op_tag_decide": (_I, [_P, _L, _I, _I, _F, _I, _P, _P, _P, _P]),
    "wfl_host_median_filter": (_I, [_P, _I, _I, _P]),
    "wfl_host_decode_bio": (_I, [_P, _I, _P, _I, _P, _P, _I, C.c_double, _P, _P, _P, _I]),
    "wfl_host_merge_segments": (_I, [_P, _P, _P, _I, _I]),
    "wfl_host_format_lab": (_L, [_P, _P, _P, _I, _P, _I, _P, _L]),
    "wfl_host_load_wav": (_I, [C.c_char_p, _P, _L, _P, _P]),
    "wfl_host_load_wavs": (_I, [_P, _I, _P, _L, _L, _P, _P, _P, _I]),
    "wfl_host_load_wav_chunks": (_I, [C.c_char_p, _I, _L, _P, _L, _I, _P, _P, _P]),
    "wfl_host_read_pcm16": (_I, [_P, _I, _P, _L, _L, _P, _P, _P, _P, _I]),
    "wfl_resample_workspace_bytes": (_L, [_I, _I]),
    "wfl_resample_pcm16": (_I, [_P, _L, _P, _P, _I, _I, _I, _P, _L, _I, _P, _L, _P]),
    "wfl_boundary_workspace_bytes": (_L, [_I, _I]),
    "wfl_boundary_features": (_I, [_P, _L, _P, _I, _I, _P, _P, _P, _P, _P, _L, _P]),
    "wfl_gemm_profile_enable": (_I, [_P, _I]),
    "wfl_gemm_profile_read": (_I, [_P, _I, _P, _P, _P, _P, _P, _I]),
}

_lib = None


class WflError(RuntimeError):
    pass


def load():
    """dlopen the in-tree library (import torch first so both share one HIP runtime)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise WflError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)      # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    if lib.wfl_abi_version() != ABI_VERSION:
        raise WflError("libwfl_asr_hip.so ABI version mismatch; rebuild")
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().wfl_last_error()
        raise WflError(f"{what}: {msg.decode() if msg else rc}")

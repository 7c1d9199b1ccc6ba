"""CPU (-m "not gpu"): the C-ABI library builds for gfx950, loads, and exports exactly what include/wfl_asr.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def libpath():
    import __graft_entry__ as g
    g.build()
    from wfl_asr_amd import _lib
    assert os.path.exists(_lib.LIB_PATH)
    return _lib.LIB_PATH


def _declared():
    src = open(os.path.join(ROOT, "include", "wfl_asr.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(wfl_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported(libpath):
    lib = ctypes.CDLL(libpath)
    names = _declared()
    assert "wfl_forward" in names and "wfl_op_gemm" in names and len(names) >= 15
    for n in names:
        assert hasattr(lib, n), n


def test_binding_table_matches_header(libpath):
    from wfl_asr_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()
    lib = _lib.load()
    assert lib.wfl_abi_version() == _lib.ABI_VERSION
    assert ctypes.sizeof(_lib.WflArch) == 64 * 4


def test_create_validates_arch_without_gpu(libpath):
    from wfl_asr_amd import _lib
    lib = _lib.load()
    a = _lib.WflArch()
    h = ctypes.c_void_p(0)
    assert lib.wfl_create(ctypes.byref(a), ctypes.byref(h)) != 0          # abi_version 0
    assert b"ABI" in lib.wfl_last_error()
    a.abi_version = _lib.ABI_VERSION
    a.d_model, a.enc_heads, a.num_classes = 100, 4, 5
    assert lib.wfl_create(ctypes.byref(a), ctypes.byref(h)) != 0
    assert b"d_model" in lib.wfl_last_error()
    a.d_model, a.enc_layers, a.enc_ffn, a.n_mels, a.max_positions = 64, 1, 128, 80, 100
    a.num_classes, a.o_id = 5, 4
    assert lib.wfl_create(ctypes.byref(a), ctypes.byref(h)) == 0 and h.value
    assert lib.wfl_num_frames(h, 12345) == 100
    assert lib.wfl_workspace_bytes(h, 2, 32000) > 0
    lib.wfl_destroy(h)


def test_precision_high_sizes_the_twin_workspace_without_gpu(libpath):
    """wfl_arch.precision (config.yaml `model.precision: high`): the activation area gets its twin for the low halves and the fp32
    accumulator of the three-pass GEMMs -- visible in wfl_workspace_bytes before any GPU is touched."""
    from wfl_asr_amd import _lib
    lib = _lib.load()
    sizes = []
    for prec in (0, 1):
        a = _lib.WflArch()
        a.abi_version = _lib.ABI_VERSION
        a.d_model, a.enc_layers, a.enc_heads, a.enc_ffn, a.n_mels, a.max_positions = 64, 1, 4, 128, 80, 100
        a.num_classes, a.o_id = 5, 4
        a.precision = prec
        h = ctypes.c_void_p(0)
        assert lib.wfl_create(ctypes.byref(a), ctypes.byref(h)) == 0 and h.value, lib.wfl_last_error()
        sizes.append(lib.wfl_workspace_bytes(h, 2, 32000))
        lib.wfl_destroy(h)
    assert sizes[0] > 0 and sizes[1] > 2 * sizes[0] * 0.9, sizes


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "wfl-asr_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                # nor the generator of synthetic checkpoints / clips (round 4: synthetic/ is test and bench infrastructure)
                assert not re.search(r"^\s*(from|import)\s+synthetic\b", txt, flags=re.M), f


def test_no_object_holds_the_packed_f32_form_that_fails_beside_mfma_waves(libpath, tmp_path):
    """gfx950: v_pk_{fma,mul,add}_f32 with op_sel[1] = 1 (the low lane takes src1's high half) returns a wrong low half in lanes 48-63
    now and then while another wave of the SIMD issues MFMAs (round 3: tools/micro/conv0_probe.hip; the cause of round 2's conv0
    corruption).  build.py checks the device assembly of every object; here: every object of the library passes, and the checker does
    catch the form."""
    from wfl_asr_amd import build as B
    objs = [os.path.join(B.OBJ, os.path.basename(s)[:-4] + ".o") for s in B.sources()]
    assert len(objs) >= 10
    for o in objs:
        B.check_device_asm(o)
    bad = tmp_path / f"x-hip-amdgcn-amd-amdhsa-{B.ARCH}.s"
    bad.write_text("\tv_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel_hi:[1,0,1]\n"
                   "\tv_pk_mov_b32 v[0:1], v[2:3], v[4:5] op_sel:[1,0]\n"
                   "\tv_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[6:7] op_sel:[1,0,0]\n")
    B.check_device_asm(str(tmp_path / "x.o"))                      # the safe selections pass
    for line in ("\tv_pk_fma_f32 v[34:35], v[34:35], v[50:51], 0 op_sel:[0,1,0] op_sel_hi:[1,1,0]\n",
                 "\tv_pk_mul_f32 v[0:1], v[2:3], v[4:5] op_sel:[0,1]\n", "\tv_pk_add_f32 v[0:1], v[2:3], v[4:5] op_sel:[1,1] op_sel_hi:[0,1]\n"):
        bad.write_text(line)
        with pytest.raises(RuntimeError, match="op_sel"):
            B.check_device_asm(str(tmp_path / "x.o"))


def test_precision_high_refuses_fp8_weights(libpath):
    """`model.precision: high` with `model.weight_dtype: fp8` is a contradiction the constructor names (no GPU needed)."""
    import pytest
    import synthetic as synth
    from wfl_asr_amd.tagger import BIOPhonemeTagger
    cfg = synth.baseline_config(4)
    cfg["model"]["precision"] = "high"
    with pytest.raises(ValueError, match="contradict"):
        BIOPhonemeTagger(cfg, synth.make_labels(5))
    cfg["model"]["weight_dtype"] = "bf16"
    BIOPhonemeTagger(cfg, synth.make_labels(5))


def test_precision_high_and_activation_formats_are_validated(libpath):
    """Advisor (round 3): `model.precision: high` silently kept the bf16 recurrence for BiLSTMs wider than 256 per direction (cfg3's
    H = 512, cfg4's H = 384).  Round 4 built the three-pass recurrence for them (csrc/lstm.hip, the four-wave form); what is left to refuse
    is a hidden size beyond 640, and the constructor says what it built."""
    import pytest
    import synthetic as synth
    from wfl_asr_amd.tagger import BIOPhonemeTagger
    labels = synth.make_labels(5)
    cfg = synth.baseline_config(3)                      # Whisper-small + full head: BiLSTM hidden 384
    assert BIOPhonemeTagger(cfg, labels).effective_precision().startswith("default")
    cfg["model"]["precision"] = "high"
    assert BIOPhonemeTagger(cfg, labels).effective_precision().startswith("high (every product")
    wide = synth.baseline_config(3)
    wide["model"].update(whisper_model="local/too-wide", precision="high")
    wide["model"]["encoder_arch"] = dict(d_model=1536, layers=2, heads=24, ffn=6144, n_mels=80, max_positions=1500)
    with pytest.raises(ValueError, match="640"):
        BIOPhonemeTagger(wide, labels)
    # activation formats of an fp8-weight model: names are checked, fp8 activations need fp8 weights
    c4 = synth.baseline_config(4)
    for name in ("bf16", "fp8", "fp8_pair", "fp8_nonscaled"):
        c4["model"]["activation_dtype"] = name
        BIOPhonemeTagger(c4, labels)
    c4["model"]["activation_dtype"] = "int8"
    with pytest.raises(ValueError, match="activation_dtype"):
        BIOPhonemeTagger(c4, labels)
    c1 = synth.baseline_config(1)
    c1["model"]["activation_dtype"] = "fp8"
    with pytest.raises(ValueError, match="weight_dtype"):
        BIOPhonemeTagger(c1, labels)


def test_gemm_split_validates_its_arguments_without_gpu(libpath):
    """wfl_op_gemm_split (the exact-label Linear / Conv1d at the operator level): argument errors come back as a status and a
    message before anything is launched."""
    from wfl_asr_amd import _lib
    lib = _lib.load()
    P = ctypes.c_void_p
    buf = (ctypes.c_char * 64)()
    a = ctypes.cast(buf, P)
    # a null operand
    rc = lib.wfl_op_gemm_split(None, a, 512, 0, 0, a, 16, 256, 512, 256, 16, 8, a, a, 256, 0, 16, None, None, None, 0, 1.0, 0, 0, None)
    assert rc != 0 and b"wfl_op_gemm_split" in lib.wfl_last_error()
    # K that is not a whole number of taps
    rc = lib.wfl_op_gemm_split(a, a, 512, 96, 96, a, 16, 256, 512, 256, 16, 8, a, a, 256, 0, 16, None, None, None, 0, 1.0, 0, 0, None)
    assert rc != 0 and b"taps" in lib.wfl_last_error()

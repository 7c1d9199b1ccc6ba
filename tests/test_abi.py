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


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "wfl-asr_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f

"""The C-ABI library builds, loads (no GPU needed) and exports exactly what include/wseg.h declares."""
import ctypes
import os
import re

from conftest import ROOT


def declared_symbols():
    with open(os.path.join(ROOT, "include", "wseg.h")) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return sorted(set(re.findall(r"\b(wseg_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from whisperseg_amd import _lib, build
    build.build(verbose=False)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), n
    assert set(names) == set(_lib.SYMBOLS), set(names) ^ set(_lib.SYMBOLS)
    bound = _lib.load()
    assert bound.wseg_abi_version() == 1


def test_struct_layouts_match_header():
    from whisperseg_amd import _lib
    assert ctypes.sizeof(_lib.ModelConfig) == 11 * 4
    assert ctypes.sizeof(_lib.LogmelDesc) == 4 * 4 + 6 * 8
    # prompt[8] + 6 ints/floats, two (pointer, int) pairs with natural alignment
    assert ctypes.sizeof(_lib.GenerateParams) == 8 * 4 + 6 * 4 + 8 + 8 + 8 + 8


def test_product_fails_loudly_without_device():
    """No CPU fallback: constructing a segmenter for 'cpu' raises instead of running something else."""
    import pytest
    from whisperseg_amd import _lib
    from whisperseg_amd.model import WhisperSegmenter
    with pytest.raises(_lib.WsegError):
        WhisperSegmenter(os.path.join(ROOT, "tests", "golden", "tiny_model"), device="cpu")

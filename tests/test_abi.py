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
    assert bound.wseg_abi_version() == 4


def test_struct_layouts_match_header(tmp_path):
    """ctypes mirrors == what a C compiler makes of include/wseg.h (sizes and every field offset)."""
    import subprocess
    from whisperseg_amd import _lib
    structs = {"wseg_model_config": _lib.ModelConfig, "wseg_logmel_desc": _lib.LogmelDesc,
               "wseg_generate_params": _lib.GenerateParams, "wseg_generate_stats": _lib.GenerateStats}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "wseg.h"', "int main(void) {"]
    for cname, ct in structs.items():
        lines.append(f'  printf("{cname} size %zu\\n", sizeof({cname}));')
        for fname, _ in ct._fields_:
            lines.append(f'  printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = {}
    for line in subprocess.check_output([str(exe)], text=True).splitlines():
        cname, field, val = line.split()
        got[(cname, field)] = int(val)
    for cname, ct in structs.items():
        assert got[(cname, "size")] == ctypes.sizeof(ct), cname
        for fname, _ in ct._fields_:
            assert got[(cname, fname)] == getattr(ct, fname).offset, (cname, fname)


def test_product_fails_loudly_without_device():
    """No CPU fallback: constructing a segmenter for 'cpu' raises instead of running something else."""
    import pytest
    from whisperseg_amd import _lib
    from whisperseg_amd.model import WhisperSegmenter
    with pytest.raises(_lib.WsegError):
        WhisperSegmenter(os.path.join(ROOT, "tests", "golden", "tiny_model"), device="cpu")


def test_fast_class_rejects_ct2_directory(tmp_path):
    """A CTranslate2-converted directory (hf_model/ with config + tokenizer but no HF weights) cannot be read; the
    constructor raises so that scripts/segment.py's try-Fast-then-fallback idiom behaves as upstream."""
    import json
    import pytest
    from whisperseg_amd.model import WhisperSegmenterFast, resolve_model_dir
    hf = tmp_path / "hf_model"
    hf.mkdir()
    (hf / "config.json").write_text(json.dumps({"total_spec_columns": 1000}))
    (tmp_path / "model.bin").write_bytes(b"ct2")
    assert resolve_model_dir(str(tmp_path)) == str(hf)
    with pytest.raises(FileNotFoundError):
        WhisperSegmenterFast(str(tmp_path), device="cuda")
    with pytest.raises(FileNotFoundError):
        resolve_model_dir(str(tmp_path / "nope"))

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
    assert bound.wseg_abi_version() == 5


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


def test_workspace_sizing_is_host_arithmetic_and_fits_the_default_api_call():
    """with paged self-attention K / V the workspace of the engine's default 1 024 slots at the reference's
    default max_length = 448 (model.py:406-409), 4 beams, whisperseg-large in the split-precision mode must fit 80 % of an
    MI355X — of the 309 220 868 096 bytes (288 GiB) the driver reports as torch's total_memory, which is what engine.pick_slots
    budgets against (measured r06; until the x3 modes' cross K / V rows grew from 192 to 196 bytes the default mode also fitted 80 %
    of 288e9: 221.8 + 6.2 GB; now 224.4 + 6.2).  wseg_workspace_bytes* are pure host arithmetic (no device needed)."""
    import ctypes as C
    from whisperseg_amd import _lib
    lib = _lib.load()
    out = {}
    for name, dtype in (("bf16", 1), ("f16x3", 4), ("f16m6", 5)):
        cfg = _lib.ModelConfig(d_model=1280, n_heads=20, enc_layers=32, dec_layers=32, ffn=5120, vocab=51865, n_mels=80,
                               spec_cols=1000, enc_positions=500, dec_positions=448, dtype=dtype)
        h = C.c_void_p()
        assert lib.wseg_model_create(C.byref(cfg), C.byref(h)) == 0
        try:
            ws = lambda s, nb, L, per=0: lib.wseg_workspace_bytes_kv(h, s, nb, L, per)
            assert lib.wseg_workspace_bytes(h, 1024, 4, 448) == ws(1024, 4, 448) == ws(1024, 4, 448, 64)
            assert 0 <= ws(1024, 4, 448) - ws(1024, 4, 64) < 64e6          # the same pool; only the bookkeeping tables follow max_length
            assert ws(1024, 4, 448, 448) > ws(1024, 4, 448) > ws(1024, 4, 35) > ws(512, 4, 35)
            assert ws(1024, 4, 448, 9999) == ws(1024, 4, 448, 448)
            assert ws(0, 4, 448) == 0 and ws(4, 9, 448) == 0 and ws(4, 4, 0) == 0 and ws(4, 4, 448, -1) == 0
            out[name] = (ws(1024, 4, 448), ws(1024, 4, 448, 448), ws(1024, 4, 35))
        finally:
            lib.wseg_model_destroy(h)
    weights_x3 = 6.2e9
    hbm = 309220868096                                                # torch.cuda.get_device_properties(0).total_memory on the MI355X boxes
    assert out["f16x3"][0] + weights_x3 <= 0.8 * hbm, out            # the API default call keeps 1 024 slots in the split modes
    assert out["f16x3"][0] + weights_x3 <= 0.8 * 288e9 + 0.5e9, out  # ... within 0.5 GB of fitting a device that reported 288e9 exactly
    assert out["f16m6"][0] + weights_x3 + 0.3e9 <= 0.8 * 288e9, out  # ... and in f16m6 (M6 operand scratch, fp32 embedding copy)
    assert out["f16x3"][1] > 288e9                                    # ... which a fully provisioned cache could never do
    assert out["bf16"][0] < 0.5 * 288e9

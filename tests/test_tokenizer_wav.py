import io
import os
import struct

import numpy as np

from tools import tiny_model as TM
from whisperseg_amd.tokenizer import WhisperSegTokenizer
from whisperseg_amd.wavio import load_wav


def test_tokenizer_roundtrip(golden_dir):
    tok = WhisperSegTokenizer.from_pretrained(os.path.join(golden_dir, "tiny_model"))
    assert tok.convert_tokens_to_ids(["<|startoftranscript|>", "<|en|>", "<|notimestamps|>"]) == TM.PROMPT
    assert tok.eos_token_id == TM.EOT == tok.pad_token_id
    ids = TM.PROMPT + TM.label_tokens([(0.24, 1.0, 2), (3.0, 3.5, 0)])
    text = tok.batch_decode([ids])[0]
    assert text == "<|startoftranscript|><|en|><|notimestamps|><|unknown|><|12|>2<|50|><|150|>0<|175|><|endoftext|>"
    assert tok.decode([TM.TIME0 + 7, 16, 15, TM.TIME0 + 9]) == "<|7|>10<|9|>"       # multi-digit cluster id
    assert tok.decode(ids, skip_special_tokens=True) == "20"


def _wav(fmt_tag, bits, channels, sr, payload):
    block = channels * bits // 8
    hdr = b"RIFF" + struct.pack("<I", 36 + len(payload)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, fmt_tag, channels, sr, sr * block, block, bits)
    return hdr + b"data" + struct.pack("<I", len(payload)) + payload


def test_wav_formats(golden_dir):
    x, sr = load_wav(os.path.join(golden_dir, "meerkat_5s.wav"))
    assert sr == 16000 and x.shape == (80000,) and x.dtype == np.float32 and 0 < np.abs(x).max() <= 1
    pcm = np.array([0, 16384, -32768, 32767], "<i2")
    y, sr = load_wav(io.BytesIO(_wav(1, 16, 1, 8000, pcm.tobytes())))
    assert sr == 8000 and y.tolist() == [0.0, 0.5, -1.0, 32767 / 32768]
    st = np.array([[1000, 3000], [-2000, 2000]], "<i2")
    y, _ = load_wav(io.BytesIO(_wav(1, 16, 2, 8000, st.tobytes())))
    assert np.allclose(y, [2000 / 32768, 0.0])
    f = np.array([0.25, -0.5], "<f4")
    y, _ = load_wav(io.BytesIO(_wav(3, 32, 1, 48000, f.tobytes())))
    assert y.tolist() == [0.25, -0.5]
    b24 = bytes([0, 0, 0x40, 0, 0, 0xC0])
    y, _ = load_wav(io.BytesIO(_wav(1, 24, 1, 48000, b24)))
    assert y.tolist() == [0.5, -0.5]


def test_tokenizer_equals_huggingface_on_both_layouts(golden_dir):
    """a-8 pinned on HF: `batch_decode(ids, skip_special_tokens=False)` of transformers' WhisperTokenizer (recorded by
    tools/make_tok_fixture.py: synthetic byte-level BPE with merged multi-digit tokens, <|0|>..<|1000|> and species tokens added the
    way reference model.py:111-113 adds them, saved the way model.py:66 saves them) against WhisperSegTokenizer reading the saved
    directory — the tokenizer.json layout transformers 5.15 writes and the vocab.json + added_tokens.json layout of 4.38.2."""
    import json
    root = os.path.join(golden_dir, "tok_fixture")
    with open(os.path.join(root, "decode_cases.json"), encoding="utf-8") as f:
        cases = json.load(f)
    assert len(cases["rows"]) >= 200
    for layout in ("hf", "slow"):
        tok = WhisperSegTokenizer.from_pretrained(os.path.join(root, layout), language="english")
        assert tok.convert_tokens_to_ids(["<|startoftranscript|>", "<|en|>", "<|notimestamps|>"]) == cases["prompt"], layout
        assert tok.eos_token_id == cases["eos_token_id"] == tok.pad_token_id
        assert tok.batch_decode(cases["rows"], skip_special_tokens=False) == cases["decoded"], layout
        assert tok.batch_decode(cases["rows"], skip_special_tokens=True) == cases["decoded_skip_special"], layout
    # the segment grammar survives: multi-digit cluster tokens sit between two time tokens with nothing inserted
    from whisperseg_amd import postprocess
    segs = postprocess.extract_segments(cases["decoded"][0], 0.01, {str(i): i for i in range(100000)})
    assert len(segs) >= 1 and all(len(r) == 3 for r in segs)


def test_tok_fixture_is_byte_reproducible(golden_dir, tmp_path):
    """A pinned fixture must be reproducible from its committed generator: tools/make_tok_fixture.py run
    here (transformers is a wheel of the image) must rewrite tests/golden/tok_fixture byte for byte."""
    import filecmp
    import subprocess
    import sys
    import pytest
    pytest.importorskip("transformers")
    from conftest import ROOT
    env = dict(os.environ, WSEG_TOK_FIXTURE_OUT=str(tmp_path), HF_HUB_OFFLINE="1")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_tok_fixture.py")], check=True, env=env, capture_output=True)
    root = os.path.join(golden_dir, "tok_fixture")
    names = ["decode_cases.json", "slow/vocab.json", "slow/added_tokens.json", "slow/merges.txt", "slow/special_tokens_map.json",
             "hf/tokenizer.json", "hf/tokenizer_config.json"]
    for name in names:
        assert filecmp.cmp(os.path.join(root, name), os.path.join(str(tmp_path), name), shallow=False), name

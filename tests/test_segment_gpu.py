"""End to end on the GPU: the shipped WhisperSegmenter / WhisperSegmenterForEval / CLI on the tiny trained
model vs the reference's own segment() outputs (tests/golden/tiny_generate.json, recorded by driving HF
fp32 through the reference's WhisperSegmenterForEval).

Bar (north star): cluster labels bit-exact, boundaries within +-1 mel frame (= spec_time_step seconds).
f32 mode is additionally required to reproduce the reference's token ids exactly."""
import io
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import golden_inputs as GI
from conftest import GOLDEN, ROOT
from tools import tiny_model as TM

pytestmark = pytest.mark.gpu
MODEL_DIR = os.path.join(GOLDEN, "tiny_model")


@pytest.fixture(scope="module")
def runs():
    with open(os.path.join(GOLDEN, "tiny_generate.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module", params=["f32", "bf16x3", "f16x3", "f16m6", "f16", "bf16"])
def segmenter(request, gpu_lib):
    from whisperseg_amd.model import WhisperSegmenter
    return request.param, WhisperSegmenter(MODEL_DIR, device="cuda", device_ids=[0], dtype=request.param)


def content(tokens):
    toks = [int(t) for t in tokens]
    if toks[:3] == TM.PROMPT:
        toks = toks[3:]
    return toks[: toks.index(TM.EOT)] if TM.EOT in toks else toks


# 16-bit modes: runs (of the 13 recorded) allowed to carry a boundary more than one mel frame off.  A time token is two mel
# frames, so any flipped time token shows as >= 2 frames; the rates over 200 recordings are in tests/test_parity_sweep_gpu.py
# (characterisation of the plain 16-bit modes; the split-precision modes — the segmenter's default — get no allowance)
BEYOND_ONE_FRAME_ALLOWED = {"f32": 0, "bf16x3": 0, "f16x3": 0, "f16m6": 0, "f16": 1, "bf16": 2}


def test_segment_matches_reference(segmenter, runs):
    dtype, seg = segmenter
    n_rows, beyond = 0, []
    for run in runs:
        audio = GI.tiny_recording(run["seed"], run["n_windows"])
        got = seg.segment(audio, TM.SR, **run["kwargs"])
        want = run["expected"]
        assert got["cluster"] == want["cluster"], (dtype, run["kwargs"])
        tol = TM.STS + 1e-9
        dev = np.abs(np.array(got["onset"] + got["offset"]) - np.array(want["onset"] + want["offset"]))
        if np.any(dev > tol):
            beyond.append((run["kwargs"], float(dev.max())))
        if dtype == "f32":
            assert got == want, run["kwargs"]          # exact rows in the exact-parity mode
        n_rows += len(want["onset"])
    assert n_rows >= 50
    assert len(beyond) <= BEYOND_ONE_FRAME_ALLOWED[dtype], (dtype, beyond)


def test_tokens_match_reference_f32(gpu_lib, runs):
    from whisperseg_amd.model import WhisperSegmenterForEval
    seg = WhisperSegmenterForEval(model_path=MODEL_DIR, device="cuda", dtype="f32")
    prompt = seg.tokenizer.convert_tokens_to_ids(["<|startoftranscript|>", "<|en|>", "<|notimestamps|>"])
    import torch
    for run in runs:
        kw = run["kwargs"]
        audio = GI.tiny_recording(run["seed"], run["n_windows"])
        sliced = seg.get_sliced_audios_features(audio, TM.SR, 0, TM.STS, kw.get("num_trials", 1))
        pos = 0
        for batch in run["token_batches"]:
            feats = torch.stack([s[2] for s in sliced[pos:pos + len(batch)]])
            toks, lens = seg.model.generate(feats, prompt, TM.EOT, TM.EOT, max_length=kw.get("max_length", 448),
                                            num_beams=kw["num_beams"], suppress_tokens=seg.suppress_tokens,
                                            begin_suppress_tokens=seg.begin_suppress_tokens)
            toks, lens = toks.cpu().numpy(), lens.cpu().numpy()
            for want, row, ln in zip(batch, toks, lens):
                assert content(row[:ln]) == content(want), (kw, pos)
            pos += len(batch)


def test_status_monitor_and_batching(segmenter):
    """Per-window results do not depend on the batch composition; progress reaches 100 (model.py:672-674)."""
    dtype, seg = segmenter
    audio = GI.tiny_recording(100, 3)
    mon = {"progress": 0}
    a = seg.segment(audio, TM.SR, batch_size=1, status_monitor=mon)
    assert mon["progress"] == 100
    b = seg.segment(audio, TM.SR, batch_size=8)
    assert a == b


def test_cli_counterpart(gpu_lib, tmp_path):
    """scripts/segment.py: same flags / CSV as the reference CLI (BASELINE config 1 plumbing, 5 s 16 kHz clip)."""
    wav = os.path.join(GOLDEN, "meerkat_5s.wav")
    out = tmp_path / "out.csv"
    env = dict(os.environ, WHISPERSEG_AMD_DTYPE="f32")
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "segment.py"), "--model_path", MODEL_DIR, "--audio_path", wav,
           "--csv_save_path", str(out), "--spec_time_step", "0.01", "--batch_size", "2"]
    subprocess.check_call(cmd, env=env)
    lines = out.read_text().strip().splitlines()
    assert lines[0] == "onset,offset,cluster"
    for ln in lines[1:]:
        on, off, c = ln.split(",")
        assert 0 <= float(on) <= float(off) <= 5.0 and c in TM.CLUSTER_CODEBOOK
    # stdin -> stdout buffer mode
    with open(wav, "rb") as f:
        res = subprocess.run(cmd[:5] + ["-", "--csv_save_path", "buffer", "--spec_time_step", "0.01"], input=f.read(),
                             env=env, capture_output=True, check=True)
    assert res.stdout.decode().strip().splitlines() == lines
    # folder mode
    folder_csv = tmp_path / "folder.csv"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "segment.py"), "--model_path", MODEL_DIR,
                           "--audio_folder", GOLDEN, "--csv_save_path", str(folder_csv), "--spec_time_step", "0.01"], env=env)
    flines = folder_csv.read_text().strip().splitlines()
    assert flines[0] == "filename,onset,offset,cluster"
    assert [ln.split(",", 1)[1] for ln in flines[1:]] == lines[1:]


def test_full_length_decode_matches_oracle(gpu_lib):
    """Maximum size: EOS suppressed so every window runs to max_length = 448 (445 steps, cache capacity, hipGraph
    replay of every step) — token-exact vs the oracle in f32 mode."""
    import torch
    from oracle import frontend as OF
    from oracle import whisper_ref as OW
    from safetensors.torch import load_file
    from whisperseg_amd.engine import Engine
    sd = {k: v.float() for k, v in load_file(os.path.join(MODEL_DIR, "model.safetensors")).items()}
    with open(os.path.join(MODEL_DIR, "config.json")) as f:
        cfg = json.load(f)
    audio = GI.tiny_recording(100, 2, tail=1.0)
    feats = torch.from_numpy(np.stack([s[2] for s in OF.sliced_audio_features(audio, TM.SR, 0, TM.STS, 1)]))
    sup = TM.SUPPRESS + [TM.EOT]
    gp = OW.GenParams(prompt=TM.PROMPT, eos_token_id=TM.EOT, pad_token_id=TM.EOT, max_length=448, num_beams=4,
                      suppress_tokens=sup, begin_suppress_tokens=TM.BEGIN_SUPPRESS)
    want = OW.generate(sd, OW.RefConfig.from_hf_dict(cfg), feats, gp)
    eng = Engine.from_state_dict(sd, cfg, "cuda:0", "f32")
    toks, lens = eng.generate(feats.cuda(), TM.PROMPT, TM.EOT, TM.EOT, max_length=448, num_beams=4, suppress_tokens=sup,
                              begin_suppress_tokens=TM.BEGIN_SUPPRESS)
    toks, lens = toks.cpu().numpy(), lens.cpu().numpy()
    assert lens.tolist() == [448, 448] and want.shape[1] == 448
    sup_set = set(sup)
    for i in range(2):
        got, ref = toks[i].tolist(), want[i].tolist()
        assert not (set(got[3:]) & sup_set)               # suppressed ids never appear
        # Forced past its natural end the tiny model is out of distribution and adjacent time tokens become near
        # ties (fp32 accumulation order then decides); require a long exact prefix rather than all 445 tokens.
        prefix = next((k for k, (a, b) in enumerate(zip(got, ref)) if a != b), 448)
        assert prefix >= 100, (i, prefix)


def test_determinism_and_batch_invariance(gpu_lib):
    """Same call twice -> identical ids (bf16 and f32); f32 mode is also invariant to the batch a window is decoded in
    (bf16 split-K plans depend on the row count, so bit-equality across batch sizes is only promised for f32)."""
    import torch
    from whisperseg_amd.model import WhisperSegmenterForEval
    for dtype in ("bf16", "f16", "f32"):
        seg = WhisperSegmenterForEval(model_path=MODEL_DIR, device="cuda", dtype=dtype)
        audio = GI.tiny_recording(102, 4)
        sliced = seg.get_sliced_audios_features(audio, TM.SR, 0, TM.STS, 1)
        feats = torch.stack([s[2] for s in sliced])
        args = dict(max_length=448, num_beams=4, suppress_tokens=seg.suppress_tokens, begin_suppress_tokens=seg.begin_suppress_tokens)
        a, la = seg.model.generate(feats, TM.PROMPT, TM.EOT, TM.EOT, **args)
        b, lb = seg.model.generate(feats, TM.PROMPT, TM.EOT, TM.EOT, **args)
        assert torch.equal(a, b) and torch.equal(la, lb)
        if dtype == "f32":
            for i in range(feats.shape[0]):
                c, lc = seg.model.generate(feats[i:i + 1], TM.PROMPT, TM.EOT, TM.EOT, **args)
                n = int(lc[0])
                assert int(la[i]) == n and torch.equal(a[i, :n], c[0, :n]), i


def test_segment_batch_pools_files(gpu_lib):
    """Continuous batching across recordings: pooled decode == per-file segment() (exact in f32 mode)."""
    from whisperseg_amd.model import WhisperSegmenter
    seg = WhisperSegmenter(MODEL_DIR, device="cuda", device_ids=[0], dtype="f32")
    audios = [GI.tiny_recording(100, 3), GI.tiny_recording(103, 1), np.zeros(0, np.float32), GI.tiny_recording(105, 2)]
    for kw in (dict(batch_size=3), dict(batch_size=5, num_trials=3), dict(batch_size=2, num_beams=1)):
        single = [seg.segment(a, TM.SR, **kw) for a in audios]
        pooled = seg.segment_batch(audios, TM.SR, **kw)
        assert pooled == single, kw
    assert sum(len(p["onset"]) for p in pooled) > 5


def test_for_eval_accepts_in_memory_hf_style_model(gpu_lib, runs):
    """reference model.py:573-601: WhisperSegmenterForEval(model=..., tokenizer=...) with an object exposing
    .state_dict() / .config / .generation_config (what train.py hands over) — weights are converted on the fly."""
    import types
    import torch
    from safetensors.torch import load_file
    from whisperseg_amd.model import WhisperSegmenterForEval
    from whisperseg_amd.tokenizer import WhisperSegTokenizer
    with open(os.path.join(MODEL_DIR, "config.json")) as f:
        cfg = json.load(f)
    sd = load_file(os.path.join(MODEL_DIR, "model.safetensors"))
    fake = types.SimpleNamespace(state_dict=lambda: {k: torch.nn.Parameter(v.float(), requires_grad=False) for k, v in sd.items()},
                                 config=cfg, generation_config=types.SimpleNamespace(suppress_tokens=TM.SUPPRESS,
                                                                                     begin_suppress_tokens=TM.BEGIN_SUPPRESS))
    seg = WhisperSegmenterForEval(model=fake, tokenizer=WhisperSegTokenizer.from_pretrained(MODEL_DIR), dtype="f32")
    run = runs[1]
    got = seg.segment(GI.tiny_recording(run["seed"], run["n_windows"]), TM.SR, **run["kwargs"])
    assert got == run["expected"]
    seg.update_cluster_codebook({"x": 0, "y": 1, "z": 2})
    renamed = seg.segment(GI.tiny_recording(run["seed"], run["n_windows"]), TM.SR, **run["kwargs"])
    assert renamed["cluster"] == [{"a": "x", "b": "y", "c": "z"}[c] for c in got["cluster"]]

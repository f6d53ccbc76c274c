"""Pins oracle/whisper_ref.py against outputs of HF `generate` driven through the reference's own
WhisperSegmenterForEval on the tiny trained model (tools/make_golden.py -> tests/golden/tiny_generate.*)."""
import json
import os

import numpy as np
import pytest
import torch

import golden_inputs as GI
from oracle import frontend as F
from oracle import whisper_ref as W
from tools import tiny_model as TM


@pytest.fixture(scope="module")
def tiny(golden_dir):
    from safetensors.torch import load_file
    mdir = os.path.join(golden_dir, "tiny_model")
    sd = {k: v.float() for k, v in load_file(os.path.join(mdir, "model.safetensors")).items()}
    with open(os.path.join(mdir, "config.json")) as f:
        cfg = json.load(f)
    with open(os.path.join(golden_dir, "tiny_generate.json")) as f:
        runs = json.load(f)
    return sd, W.RefConfig.from_hf_dict(cfg), runs, np.load(os.path.join(golden_dir, "tiny_generate.npz"))


def content(tokens):
    """generated ids before the first EOS, prompt stripped (HF 4.38.2 returns the prompt, 5.15 does not)."""
    c = W.canonical(tokens, 3, TM.EOT, TM.PROMPT)
    return c[:-1] if c and c[-1] == TM.EOT else c


def test_encoder_and_first_logits(tiny):
    sd, rc, runs, z = tiny
    audio = GI.tiny_recording(100, 3)
    sliced = F.sliced_audio_features(audio, TM.SR, 0, TM.STS, 1)
    feats = torch.from_numpy(np.stack([s[2] for s in sliced]))
    assert np.max(np.abs(feats.numpy()[:, :, ::GI.COL_STRIDE] - z["features_cols"])) <= 1e-4
    enc = W.encoder_forward(sd, rc, feats)
    assert np.max(np.abs(enc.numpy()[:, ::25, :] - z["enc_out_sample"])) <= 2e-4
    gp = W.GenParams(prompt=TM.PROMPT, eos_token_id=TM.EOT, pad_token_id=TM.EOT, max_length=8, num_beams=1)
    _, logits = W.generate(sd, rc, feats, gp, return_first_logits=True)
    assert np.max(np.abs(logits.numpy() - z["first_logits"])) <= 2e-3


def test_token_sequences_match_hf(tiny):
    """Greedy and beam search (beams 1/2/4, default and short max_length, EOS at varied lengths)."""
    sd, rc, runs, _ = tiny
    n_checked = 0
    for run in runs:
        kw = run["kwargs"]
        audio = GI.tiny_recording(run["seed"], run["n_windows"])
        sliced = F.sliced_audio_features(audio, TM.SR, 0, TM.STS, kw.get("num_trials", 1))
        feats = torch.from_numpy(np.stack([s[2] for s in sliced]))
        gp = W.GenParams(prompt=TM.PROMPT, eos_token_id=TM.EOT, pad_token_id=TM.EOT, max_length=kw.get("max_length", 448),
                         num_beams=kw["num_beams"], suppress_tokens=TM.SUPPRESS, begin_suppress_tokens=TM.BEGIN_SUPPRESS)
        pos = 0
        for batch in run["token_batches"]:
            out = W.generate(sd, rc, feats[pos:pos + len(batch)], gp)
            for want, got in zip(batch, out.tolist()):
                assert content(got) == content(want), (kw, pos)
                n_checked += 1
            pos += len(batch)
        assert pos == len(sliced)
    assert n_checked >= 20


def test_eos_fires_at_varied_lengths(tiny):
    _, _, runs, _ = tiny
    lengths = {len(content(row)) for run in runs for b in run["token_batches"] for row in b}
    assert len(lengths) >= 5


# ---- the second (held-out) fixture model: another geometry (d 256, 4 heads, 3 + 3 layers), full-mantissa fp32 weights -----------------
@pytest.fixture(scope="module")
def tiny2(golden_dir):
    from safetensors.torch import load_file
    mdir = os.path.join(golden_dir, "tiny_model2")
    sd = {k: v.float() for k, v in load_file(os.path.join(mdir, "model.safetensors")).items()}
    with open(os.path.join(mdir, "config.json")) as f:
        cfg = json.load(f)
    with open(os.path.join(golden_dir, "tiny2_sweep.json")) as f:
        sweep = json.load(f)
    return sd, W.RefConfig.from_hf_dict(cfg), sweep, np.load(os.path.join(golden_dir, "tiny2_generate.npz"))


def test_second_fixture_model_is_another_model(tiny, tiny2):
    sd1, rc1 = tiny[0], tiny[1]
    sd2, rc2, sweep, _ = tiny2
    assert (rc2.d_model, rc2.encoder_layers, rc2.decoder_layers) == (256, 3, 3) and rc1.d_model == 128
    w = sd2["model.decoder.layers.0.fc1.weight"]
    assert not torch.equal(w, w.half().float()) and not torch.equal(w, w.bfloat16().float())      # full-mantissa weights: lo halves are live
    assert len(sweep) == 1000 and {r["seed"] for r in sweep} == set(range(5000, 5250))


def test_second_fixture_encoder_and_first_logits(tiny2):
    """oracle vs HF (recorded by tools/make_golden.py --only sweep2) on the held-out model's geometry"""
    sd, rc, _, z = tiny2
    audio = GI.tiny_recording(5000, 3, variant="tiny2")
    sliced = F.sliced_audio_features(audio, TM.SR, 0, TM.STS, 1)
    feats = torch.from_numpy(np.stack([s[2] for s in sliced]))
    enc = W.encoder_forward(sd, rc, feats)
    assert np.max(np.abs(enc.numpy()[:, ::25, :] - z["enc_out_sample"])) <= 5e-4
    gp = W.GenParams(prompt=TM.PROMPT, eos_token_id=TM.EOT, pad_token_id=TM.EOT, max_length=8, num_beams=1)
    _, logits = W.generate(sd, rc, feats, gp, return_first_logits=True)
    assert np.max(np.abs(logits.numpy() - z["first_logits"])) <= 3e-3


def test_oracle_reproduces_heldout_rows(tiny2, golden_dir):
    """the CPU oracle + the product's host epilogue through 12 of the 1 000 held-out recordings (all four (trials, beams) combinations)"""
    from tools.precision_study import OracleSegmenter, Policy
    _, _, sweep, _ = tiny2
    seg = OracleSegmenter(Policy(""), model_dir=os.path.join(golden_dir, "tiny_model2"))
    for run in sweep[0:8] + sweep[500:504]:
        got = seg.segment(GI.tiny_recording(run["seed"], run["n_windows"], variant="tiny2"), TM.SR, **run["kwargs"])
        assert got == run["expected"], (run["seed"], run["kwargs"])


# ---- the third fixture model (d 128, 2 heads, 4 + 4 layers, ffn 640, fp32 weights) ---------------------------------------------------
def test_third_fixture_encoder_logits_and_rows(golden_dir):
    """oracle vs HF on the third model's geometry (tools/make_golden.py --only sweep5): encoder output, first-step logits, and 8 of its
    1 000 sweep recordings end to end through the oracle + the product's host epilogue"""
    from safetensors.torch import load_file
    from tools.precision_study import OracleSegmenter, Policy
    mdir = os.path.join(golden_dir, "tiny_model3")
    sd = {k: v.float() for k, v in load_file(os.path.join(mdir, "model.safetensors")).items()}
    with open(os.path.join(mdir, "config.json")) as f:
        rc = W.RefConfig.from_hf_dict(json.load(f))
    assert (rc.d_model, rc.encoder_layers, rc.decoder_layers) == (128, 4, 4)
    z = np.load(os.path.join(golden_dir, "tiny3_generate.npz"))
    with open(os.path.join(golden_dir, "tiny3_sweep.json")) as f:
        sweep = json.load(f)
    assert len(sweep) == 1000 and {r["seed"] for r in sweep} == set(range(11000, 11250))
    audio = GI.tiny_recording(11000, 3, variant="tiny3")
    sliced = F.sliced_audio_features(audio, TM.SR, 0, TM.STS, 1)
    feats = torch.from_numpy(np.stack([s[2] for s in sliced]))
    enc = W.encoder_forward(sd, rc, feats)
    assert np.max(np.abs(enc.numpy()[:, ::25, :] - z["enc_out_sample"])) <= 5e-4
    gp = W.GenParams(prompt=TM.PROMPT, eos_token_id=TM.EOT, pad_token_id=TM.EOT, max_length=8, num_beams=1)
    _, logits = W.generate(sd, rc, feats, gp, return_first_logits=True)
    assert np.max(np.abs(logits.numpy() - z["first_logits"])) <= 3e-3
    seg = OracleSegmenter(Policy(""), model_dir=mdir)
    with open(os.path.join(golden_dir, "tiny3_sweep6.json")) as f:      # sweep 6: 1 000 more of the same model (tools/record_sweep.sh sweep6)
        sweep6 = json.load(f)
    assert len(sweep6) == 1000 and {r["seed"] for r in sweep6} == set(range(13000, 13250))
    with open(os.path.join(golden_dir, "tiny3_sweep7.json")) as f:      # sweep 7: and 1 000 more (tools/record_sweep.sh sweep7)
        sweep7 = json.load(f)
    assert len(sweep7) == 1000 and {r["seed"] for r in sweep7} == set(range(15000, 15250))
    for run in sweep[0:4] + sweep[600:604] + sweep6[0:4] + sweep6[700:702] + sweep7[300:302]:
        got = seg.segment(GI.tiny_recording(run["seed"], run["n_windows"], variant="tiny3"), TM.SR, **run["kwargs"])
        assert got == run["expected"], (run["seed"], run["kwargs"])

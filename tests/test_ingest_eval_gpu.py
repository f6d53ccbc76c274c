"""Callers either side of the path on the GPU: mixed sampling rates (BASELINE configs[4]), the resampler in
front of segment_batch, the evaluation harness on a labelled folder, and sharded checkpoints on the device."""
import json
import os
import shutil
import struct

import numpy as np
import pytest
import torch

import golden_inputs as GI
from conftest import GOLDEN
from tools import tiny_model as TM

pytestmark = pytest.mark.gpu
MODEL_DIR = os.path.join(GOLDEN, "tiny_model")


def write_wav(path, x, sr, fmt="pcm16"):
    if fmt == "pcm16":
        data = (np.clip(x, -1, 1 - 1 / 32768) * 32768.0).round().astype("<i2").tobytes()
        tag, bits = 1, 16
    else:
        data = x.astype("<f4").tobytes()
        tag, bits = 3, 32
    hdr = b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, tag, 1, sr, sr * bits // 8, bits // 8, bits) \
        + b"data" + struct.pack("<I", len(data))
    with open(path, "wb") as f:
        f.write(hdr + data)


def rows_close(a, b, tol):
    return a["cluster"] == b["cluster"] and len(a["onset"]) == len(b["onset"]) and \
        all(abs(x - y) <= tol for x, y in zip(a["onset"] + a["offset"], b["onset"] + b["offset"]))


def test_mixed_rate_files_pooled_equal_per_file(gpu_lib):
    """configs[4]: recordings at 16 / 32 / 48 kHz (three front-end configurations: hop 160/320/480, n_fft 512/512/1024) are
    pooled into one decode; per-recording results equal separate segment() calls exactly in f32 mode.  (What the tiny model,
    trained on 16 kHz features only, makes of the other two spectra is irrelevant here.)"""
    from scipy.signal import resample_poly
    from whisperseg_amd.model import WhisperSegmenter
    seg = WhisperSegmenter(MODEL_DIR, device="cuda", device_ids=[0], dtype="f32")
    base = [GI.tiny_recording(100, 3), GI.tiny_recording(103, 2), GI.tiny_recording(105, 2)]
    audios = [base[0], resample_poly(base[1], 2, 1).astype(np.float32), resample_poly(base[2], 3, 1).astype(np.float32)]
    srs = [16000, 32000, 48000]
    single = [seg.segment(a, sr, spec_time_step=TM.STS) for a, sr in zip(audios, srs)]
    pooled = seg.segment_batch(audios, srs, spec_time_step=TM.STS)
    assert pooled == single
    assert sum(len(p["onset"]) for p in pooled) >= 8


def species_batch():
    """A multi-species batch in the reference's sense (config/segment_config.json:1-49): per-recording sampling rate, spectrogram
    time step, minimum frequency, trial count and post-filter lengths."""
    from scipy.signal import resample_poly
    base = [GI.tiny_recording(100, 3), GI.tiny_recording(103, 2), GI.tiny_recording(105, 2), GI.tiny_recording(107, 1)]
    audios = [base[0], resample_poly(base[1], 2, 1).astype(np.float32), resample_poly(base[2], 3, 1).astype(np.float32), base[3]]
    srs = [16000, 32000, 48000, 16000]
    kw = dict(spec_time_step=[TM.STS, 0.005, 0.0025, None], min_frequency=[0, 0, 2000, None], num_trials=[1, 3, 2, 1],
              min_segment_length=[None, 0.01, 0.0, None], eps=[None, 0.02, None, None])
    return audios, srs, kw


def per_file(seg, audios, srs, kw):
    return [seg.segment(a, sr, **{k: v[i] for k, v in kw.items()}) for i, (a, sr) in enumerate(zip(audios, srs))]


def test_multi_species_batch_is_one_pooled_decode(gpu_lib):
    """configs[4] "multi-species batch": recordings with DIFFERENT sr / spec_time_step / min_frequency / num_trials / filter
    lengths go through one segment_batch call = one pooled decode, each with its own front-end configuration; results equal
    per-file segment() with the same parameters (exact in f32 mode, and in the default split-precision mode the rows agree
    too).  Reference: evaluate.py:15-24 passes these per file; config/segment_config.json:1-49 ships them per species."""
    from whisperseg_amd.model import WhisperSegmenter
    audios, srs, kw = species_batch()
    for dtype in ("f32", None):          # None: the default mode
        seg = WhisperSegmenter(MODEL_DIR, device="cuda", device_ids=[0], dtype=dtype)
        calls = []
        inner = seg.generate_segment_text
        seg.generate_segment_text = lambda sliced, *a, **k: (calls.append(len(sliced)), inner(sliced, *a, **k))[1]
        pooled = seg.segment_batch(audios, srs, **kw)
        assert len(calls) == 1 and calls[0] >= 15        # ONE decode over the pooled windows of all four recordings
        seg.generate_segment_text = inner
        assert pooled == per_file(seg, audios, srs, kw), dtype
    assert sum(len(p["onset"]) for p in pooled) >= 8


def test_distributed_clip_batch_on_the_gpu(gpu_lib):
    """dist.segment_batch_distributed with the RCCL collectives live (WSEG_FORCE_DIST=1: a one-rank nccl group in a child
    process): metadata broadcast, PCM hand-over, token all_gather, per-recording parse == per-file segment()."""
    import subprocess
    import sys
    code = r"""
import os, sys, json
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import torch
from whisperseg_amd import dist as wd
from whisperseg_amd.model import WhisperSegmenter
import test_ingest_eval_gpu as T
rank, world, _ = wd.init_from_env()
assert torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl"
seg = WhisperSegmenter(T.MODEL_DIR, device="cuda", device_ids=[0], dtype="f32")
audios, srs, kw = T.species_batch()
got = wd.segment_batch_distributed(seg, audios, srs, **kw)
assert got == T.per_file(seg, audios, srs, kw)
one = wd.segment_distributed(seg, audios[1], srs[1], **{k: v[1] for k, v in kw.items()})
assert one == got[1]
print("DIST-OK", sum(len(p["onset"]) for p in got))
torch.distributed.destroy_process_group()
"""
    from conftest import ROOT
    env = dict(os.environ, WSEG_FORCE_DIST="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(29700 + os.getpid() % 200), HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, "-c", code, ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0 and "DIST-OK" in res.stdout, res.stderr[-3000:]


def test_resample_then_segment_batch(gpu_lib):
    """"mixed sr 16/32/48 kHz resampled": 32 / 48 kHz sources -> GPU polyphase resampler -> 16 kHz model input (device tensors
    go straight into segment_batch, no host round trip) -> the segments of the original 16 kHz recordings."""
    from scipy.signal import resample_poly
    from whisperseg_amd.model import WhisperSegmenter
    from whisperseg_amd.resample import resample
    seg = WhisperSegmenter(MODEL_DIR, device="cuda", device_ids=[0], dtype="f32")
    base = [GI.tiny_recording(100, 3), GI.tiny_recording(103, 2), GI.tiny_recording(102, 2)]
    srs = [16000, 32000, 48000]
    sources = [base[0], resample_poly(base[1], 2, 1).astype(np.float32), resample_poly(base[2], 3, 1).astype(np.float32)]
    model_in = [resample(torch.from_numpy(s).cuda(), sr, TM.SR) if sr != TM.SR else s for s, sr in zip(sources, srs)]
    assert all(len(m) == len(b) for m, b in zip(model_in, base))
    pooled = seg.segment_batch(model_in, TM.SR)
    single = [seg.segment(m, TM.SR) for m in model_in]
    assert pooled == single
    want = [seg.segment(b, TM.SR) for b in base]
    for got, ref in zip(pooled, want):
        assert rows_close(got, ref, TM.STS + 1e-9)        # up- then down-sampling is not the identity: +-1 frame


def test_evaluate_dataset_on_a_labelled_folder(gpu_lib, tmp_path):
    """reference evaluate.py:53-84 on wav + label files: labels = the rows the REFERENCE's segment() produced for the same
    recordings (tests/golden/tiny_generate.json), audio stored as float32 wav (bit-exact samples), so the f32 engine must
    score F1 = 1.0 segment-wise and frame-wise; a second folder with shifted labels must score lower."""
    from whisperseg_amd.evaluate import evaluate_dataset
    with open(os.path.join(GOLDEN, "tiny_generate.json")) as f:
        runs = [r for r in json.load(f) if r["kwargs"].get("num_trials", 1) == 1 and r["kwargs"]["num_beams"] == 4
                and "max_length" not in r["kwargs"]]
    assert runs
    good, bad = tmp_path / "good", tmp_path / "bad"
    good.mkdir(), bad.mkdir()
    n_rows = 0
    for i, run in enumerate(runs):
        audio = GI.tiny_recording(run["seed"], run["n_windows"])
        exp = run["expected"]
        n_rows += len(exp["onset"])
        for folder, shift in ((good, 0.0), (bad, 0.3)):
            write_wav(str(folder / f"rec{i}.wav"), audio, TM.SR, "float32")
            label = dict(onset=[v + shift for v in exp["onset"]], offset=[v + shift for v in exp["offset"]], cluster=exp["cluster"],
                         sr=TM.SR, spec_time_step=TM.STS, min_frequency=0)
            with open(folder / f"rec{i}.json", "w") as f:
                json.dump(label, f)
    assert n_rows >= 5
    os.environ["WHISPERSEG_AMD_DTYPE"] = "f32"
    try:
        res = evaluate_dataset(str(good), MODEL_DIR, num_trials=1)
        res_bad = evaluate_dataset(str(bad), MODEL_DIR, num_trials=1)
    finally:
        del os.environ["WHISPERSEG_AMD_DTYPE"]
    assert res["segment_wise_scores"]["N-positive-in-ground-truth"] == n_rows
    assert res["segment_wise_scores"]["F1"] == pytest.approx(1.0) and res["frame_wise_scores"]["F1"] == pytest.approx(1.0)
    assert res_bad["segment_wise_scores"]["F1"] < 0.5


def test_sharded_checkpoint_loads_on_the_device(gpu_lib, tmp_path):
    """A sharded copy of the tiny model (what save_pretrained writes for a > 5 GB fp32 whisperseg-large, reference
    model.py:59-74) gives the same engine as the single-file original; .bin shards too."""
    from safetensors.torch import load_file, save_file
    from whisperseg_amd.model import WhisperSegmenter
    sd = load_file(os.path.join(MODEL_DIR, "model.safetensors"))
    names = list(sd)
    for kind in ("safetensors", "bin"):
        dst = tmp_path / kind
        dst.mkdir()
        for fn in ("config.json", "generation_config.json", "vocab.json", "added_tokens.json"):
            shutil.copy(os.path.join(MODEL_DIR, fn), dst / fn)
        wm, per = {}, (len(names) + 2) // 3
        for i in range(3):
            part = {k: sd[k].contiguous() for k in names[i * per:(i + 1) * per]}
            fname = f"model-{i + 1:05d}-of-00003.safetensors" if kind == "safetensors" else f"pytorch_model-{i + 1:05d}-of-00003.bin"
            save_file(part, str(dst / fname)) if kind == "safetensors" else torch.save(part, str(dst / fname))
            wm.update({k: fname for k in part})
        index = "model.safetensors.index.json" if kind == "safetensors" else "pytorch_model.bin.index.json"
        with open(dst / index, "w") as f:
            json.dump({"metadata": {}, "weight_map": wm}, f)
    ref = WhisperSegmenter(MODEL_DIR, device="cuda", device_ids=[0], dtype="bf16")
    audio = GI.tiny_recording(100, 3)
    want = ref.segment(audio, TM.SR)
    for kind in ("safetensors", "bin"):
        seg = WhisperSegmenter(str(tmp_path / kind), device="cuda", device_ids=[0], dtype="bf16")
        for k, v in ref.model_list[0].weights.items():
            assert torch.equal(v, seg.model_list[0].weights[k]), (kind, k)
        assert seg.segment(audio, TM.SR) == want

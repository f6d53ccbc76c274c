"""Parity of WhisperSegmenter.segment() in every engine mode (f32, f16m6, f16x3, bf16x3, f16, bf16) against the reference's own rows over tests/golden/tiny_sweep.json:
200 recordings (50 seeds x trials {1, 3} x beams {1, 4}) of the tiny trained model, expected rows recorded by driving HF fp32
through the reference's WhisperSegmenterForEval (tools/make_golden.py, G8).

    python tools/parity_sweep.py [out.json] [modes ...]          (needs the GPU)
    python tools/parity_sweep.py --sweeps profiles/r06_parity_sweeps.json      both sweeps (r06): the 200 recordings above AND the held-out
                                 1 000 recordings of the second fixture model (tests/golden/tiny2_sweep.json, tiny_model2) in every mode;
                                 bench.py reads its `parity` strings from the committed file

Per dtype: runs whose rows have the same count and clusters as the reference's ("structure"), the histogram of boundary
deviations in mel frames (spec_time_step units) over all rows of those runs, and the list of runs outside the north-star
tolerance (clusters exact, boundaries within +-1 frame)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_inputs as GI  # noqa: E402
from tools import tiny_model as TM  # noqa: E402

MODEL_DIR = os.path.join(ROOT, "tests", "golden", "tiny_model")
# name -> (rows recorded from the reference, fixture model directory, signal family of its recordings)
SWEEPS = {"sweep1": ("tiny_sweep.json", "tiny_model", "tiny"), "sweep2_heldout": ("tiny2_sweep.json", "tiny_model2", "tiny2"),
          # 1 000 further recordings of the second model, recorded after sweep 2 had been looked at (tools/make_golden.py --only sweep3)
          "sweep3_fresh": ("tiny2_sweep3.json", "tiny_model2", "tiny2"),
          # 1 000 more, recorded after the default had moved to f16x3 (tools/record_sweep.sh sweep4)
          "sweep4_more": ("tiny2_sweep4.json", "tiny_model2", "tiny2"),
          # 1 000 recordings of a third model (d 128, 4 + 4 layers; tools/record_sweep.sh sweep5)
          "sweep5_third_model": ("tiny3_sweep.json", "tiny_model3", "tiny3"),
          # 1 000 more of the third model, recorded while the x3 modes' cross K / V moved to 24-bit block floating point — the remedy for
          # f16x3's one miss on sweep 5 — and first scored after that format was frozen (tools/record_sweep.sh sweep6)
          "sweep6_third_fresh": ("tiny3_sweep6.json", "tiny_model3", "tiny3"),
          # ... and 1 000 more, recorded after that format had been frozen and scored on sweep 6 (tools/record_sweep.sh sweep7)
          "sweep7_third_fresh2": ("tiny3_sweep7.json", "tiny_model3", "tiny3")}


def _new_tally(n):
    return dict(runs=n, rows_expected=0, rows_compared=0, exact_runs=0, within_tolerance_runs=0, structure_mismatch_runs=[],
                beyond_one_frame_runs=[], cluster_mismatch_rows=0, frame_hist={"0": 0, "<=0.5": 0, "<=1": 0, "<=2": 0, ">2": 0}, max_dev_frames=0.0)


def _tally(out, idx, run, got):
    """Score one recording's rows `got` against the reference's (run["expected"]) into the tally `out`."""
    hist = out["frame_hist"]
    want = run["expected"]
    out["rows_expected"] += len(want["onset"])
    if got == want:
        out["exact_runs"] += 1
    if len(got["onset"]) != len(want["onset"]) or got["cluster"] != want["cluster"]:
        out["structure_mismatch_runs"].append(dict(index=idx, seed=run["seed"], kwargs=run["kwargs"], got_rows=len(got["onset"]),
                                                   want_rows=len(want["onset"])))
        if len(got["onset"]) == len(want["onset"]):
            out["cluster_mismatch_rows"] += sum(a != b for a, b in zip(got["cluster"], want["cluster"]))
        return
    dev = np.abs(np.array(got["onset"] + got["offset"]) - np.array(want["onset"] + want["offset"])) / TM.STS
    out["rows_compared"] += len(want["onset"])
    for d in dev:
        k = "0" if d < 1e-6 else ("<=0.5" if d <= 0.5 + 1e-6 else ("<=1" if d <= 1 + 1e-6 else ("<=2" if d <= 2 + 1e-6 else ">2")))
        hist[k] += 1
    mx = float(dev.max()) if len(dev) else 0.0
    out["max_dev_frames"] = max(out["max_dev_frames"], mx)
    if mx <= 1 + 1e-6:
        out["within_tolerance_runs"] += 1
    else:
        out["beyond_one_frame_runs"].append(dict(index=idx, seed=run["seed"], kwargs=run["kwargs"], max_dev_frames=mx))


def score(seg, sweep, variant="tiny"):
    """Every recording through its own segment() call (what the reference's rows were recorded with)."""
    out = _new_tally(len(sweep))
    audio_cache = {}
    for idx, run in enumerate(sweep):
        key = (run["seed"], run["n_windows"])
        if key not in audio_cache:
            audio_cache[key] = GI.tiny_recording(*key, variant=variant)
        _tally(out, idx, run, seg.segment(audio_cache[key], TM.SR, **run["kwargs"]))
    return out


def score_pooled(seg, sweep, variant="tiny"):
    """The same recordings through the product's POOLED path: one segment_batch() call per beam count (the decode parameter that is per
    call), with each recording's own num_trials — all windows of ~500 recordings share the engine's slots, are admitted in whatever groups
    the scheduler forms and finish at their own lengths.  Scored exactly like score(): every row must still be the reference's."""
    out = _new_tally(len(sweep))
    audio_cache = {}
    by_beams = {}
    for idx, run in enumerate(sweep):
        by_beams.setdefault(run["kwargs"]["num_beams"], []).append(idx)
    for beams, idxs in sorted(by_beams.items()):
        audios, trials = [], []
        for idx in idxs:
            run = sweep[idx]
            key = (run["seed"], run["n_windows"])
            if key not in audio_cache:
                audio_cache[key] = GI.tiny_recording(*key, variant=variant)
            audios.append(audio_cache[key])
            trials.append(run["kwargs"]["num_trials"])
            extra = {k: v for k, v in run["kwargs"].items() if k not in ("num_beams", "num_trials", "batch_size")}
            assert not extra, extra      # the sweeps vary trials and beams only
        preds = seg.segment_batch(audios, TM.SR, num_trials=trials, num_beams=beams, batch_size=8)
        for idx, got in zip(idxs, preds):
            _tally(out, idx, sweep[idx], got)
    return out


class HybridEngine:
    """Attribution experiment: encoder of one engine, decoder of another (wseg_generate's encoder_output hook)."""

    def __init__(self, enc_engine, dec_engine):
        self.enc, self.dec = enc_engine, dec_engine
        self.device, self.geo = dec_engine.device, dec_engine.geo

    def generate(self, feats, *a, **kw):
        return self.dec.generate(feats, *a, encoder_output=self.enc.encode(feats), **kw)


def summary(r):
    """The part of a score() result that is committed (run lists cut to their first entries)."""
    out = {k: v for k, v in r.items() if not k.endswith("_runs") or isinstance(v, int)}
    out["structure_mismatch_runs"] = len(r["structure_mismatch_runs"])
    out["beyond_one_frame_runs"] = len(r["beyond_one_frame_runs"])
    out["first_bad_runs"] = (r["structure_mismatch_runs"] + r["beyond_one_frame_runs"])[:5]
    return out


def both_sweeps(dest, modes):
    from whisperseg_amd.model import WhisperSegmenter
    res = {}
    only = os.environ.get("SWEEP_ONLY")      # e.g. SWEEP_ONLY=sweep3_fresh
    for name, (rows, mdir, variant) in SWEEPS.items():
        if (only and name != only) or not os.path.exists(os.path.join(ROOT, "tests", "golden", rows)):
            continue
        with open(os.path.join(ROOT, "tests", "golden", rows)) as f:
            sweep = json.load(f)
        for dtype in modes:
            seg = WhisperSegmenter(os.path.join(ROOT, "tests", "golden", mdir), device="cuda", device_ids=[0], dtype=dtype)
            r = summary(score(seg, sweep, variant))
            res.setdefault(dtype, {})[name] = r
            print(name, dtype, "runs", r["runs"], "exact", r["exact_runs"], "within +-1 frame", r["within_tolerance_runs"], "hist", r["frame_hist"], flush=True)
            del seg
    with open(dest, "w") as f:
        json.dump(res, f, indent=1)


def main():
    from whisperseg_amd.model import WhisperSegmenter
    if len(sys.argv) > 1 and sys.argv[1] == "--sweeps":
        return both_sweeps(sys.argv[2], sys.argv[3:] or ["f32", "f16m6", "f16x3", "bf16x3", "f16", "bf16"])
    with open(os.path.join(ROOT, "tests", "golden", "tiny_sweep.json")) as f:
        sweep = json.load(f)
    res = {}
    modes = [m for m in sys.argv[2:]] or ["f32", "f16m6", "f16x3", "bf16x3", "f16", "bf16"]
    segs = {}
    for dtype in modes:
        if "+" in dtype:       # "enc:f32+dec:bf16"
            e, d = (x.split(":")[1] for x in dtype.split("+"))
            for x in (e, d):
                segs.setdefault(x, WhisperSegmenter(MODEL_DIR, device="cuda", device_ids=[0], dtype=x))
            seg = WhisperSegmenter(MODEL_DIR, device="cuda", device_ids=[0], dtype=d)
            seg.model_list = [HybridEngine(segs[e].model_list[0], segs[d].model_list[0])]
        else:
            seg = segs.setdefault(dtype, WhisperSegmenter(MODEL_DIR, device="cuda", device_ids=[0], dtype=dtype))
        res[dtype] = score(seg, sweep)
        r = res[dtype]
        print(dtype, "runs", r["runs"], "exact", r["exact_runs"], "within +-1 frame", r["within_tolerance_runs"], "structure mismatches",
              len(r["structure_mismatch_runs"]), "hist", r["frame_hist"], "max", r["max_dev_frames"], flush=True)
    dest = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "parity_sweep.json")
    os.makedirs(os.path.dirname(dest), exist_ok=True)
    with open(dest, "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()

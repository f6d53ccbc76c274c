"""North-star parity over SEVEN sweeps (6 200 recordings, three fixture models) of rows recorded from the reference's own segment() on HF
fp32 (tools/make_golden.py):
sweep 1 — 200 recordings of the first fixture model, the set every precision format of r03-r05 was chosen on; sweep 2 — 1 000 held-out
recordings of a second, independently trained model (formats frozen before it was recorded); sweep 3 — 1 000 further recordings
recorded after sweep 2 had been looked at (the fresh test of the default that sweep 2 led to); sweep 4 — 1 000 more; sweep 5 — 1 000
recordings of a THIRD model (deeper and narrower); sweep 6 — 1 000 more of the third model, first scored after the last format change
of r06; sweep 7 — 1 000 more, recorded after sweep 6 had been scored.  The exact mode f32 must reproduce EVERY row of all seven.  The product default f16x3 does too (asserted: 0 recordings beyond
+-1 mel frame, clusters bit-exact, rows identical, on all 6 200) — since its cross K / V rows became 24-bit block floating point: with
the 24-bit FLOAT rows of r03 - mid r06 it missed ONE recording of sweep 5 (a greedy decision whose top-1 / top-2 margin in the fp32
oracle is 2.8e-5; profiles/r06_parity_sweeps_x3_k24rows.json), which is what that change was made for, and sweeps 6 and 7 are the fresh
tests of it.  bf16x3 reproduces sweeps 1-4, 6 and 7 and misses two recordings of sweep 5 (its GEMM error, not the rows).  No 16-bit-operand mode is
exact by construction; the measured rates are 0 / 2 / 10 of 6 200 (f16x3 / bf16x3 / f16m6; profiles/r06_parity_sweeps.json).  f16m6 (the
default of r04-r05) reproduces sweeps 1 and 6 and is OUTSIDE the tolerance on 2 / 3 / 1 / 3 / 1 of the 1 000 recordings of sweeps 2 / 3 / 4 / 5 / 7:
it is characterised, like the plain 16-bit modes f16 / bf16 — they must stay inside the measured envelope committed in
profiles/r06_parity_sweeps.json (scored with tools/parity_sweep.py)."""
import json
import os

import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu
MODEL_DIR = os.path.join(GOLDEN, "tiny_model")


@pytest.fixture(scope="module")
def sweep():
    with open(os.path.join(GOLDEN, "tiny_sweep.json")) as f:
        return json.load(f)


def test_sweep_is_big_enough(sweep):
    assert len(sweep) == 200
    assert sum(len(r["expected"]["onset"]) for r in sweep) >= 800
    multi = [r for r in sweep if r["kwargs"]["num_trials"] == 3 and r["expected"]["onset"]]
    assert len(multi) >= 50          # multi-trial consolidation with non-empty results


def test_f32_mode_reproduces_every_row(gpu_lib, sweep):
    from tools.parity_sweep import score
    from whisperseg_amd.model import WhisperSegmenter
    res = score(WhisperSegmenter(MODEL_DIR, device="cuda", device_ids=[0], dtype="f32"), sweep)
    assert res["exact_runs"] == len(sweep), (res["structure_mismatch_runs"][:3], res["beyond_one_frame_runs"][:3])


@pytest.mark.parametrize("dtype", ["bf16x3", "f16x3", "f16m6"])
def test_split_precision_modes_meet_the_north_star_tolerance(gpu_lib, sweep, dtype):
    """Sweep 1.  The split-precision modes: GEMM operands as hi + lo 16-bit pairs, three MFMAs per product (f16x3 / bf16x3) or hi*hi on
    the half matrix cores + both cross terms on the fp6 MX matrix cores (f16m6: its formats were chosen ON this sweep, which it
    reproduces; the held-out sweeps below put it outside the tolerance), fp32 everywhere else.  North star: clusters exact, boundaries
    within +-1 mel frame — on EVERY recording of the sweep."""
    from tools.parity_sweep import score
    from whisperseg_amd.model import WhisperSegmenter
    res = score(WhisperSegmenter(MODEL_DIR, device="cuda", device_ids=[0], dtype=dtype), sweep)
    print(dtype, json.dumps({k: v for k, v in res.items() if not k.endswith("_runs") or isinstance(v, int)}))
    assert res["structure_mismatch_runs"] == [] and res["beyond_one_frame_runs"] == [], (res["structure_mismatch_runs"][:3], res["beyond_one_frame_runs"][:3])
    assert res["cluster_mismatch_rows"] == 0
    assert res["within_tolerance_runs"] == len(sweep)


@pytest.mark.parametrize("dtype", ["f16", "bf16"])
def test_16_bit_modes_stay_inside_their_measured_envelope(gpu_lib, sweep, dtype):
    """CHARACTERISATION of the plain 16-bit modes (not the tolerance: that is asserted for the split-precision modes above).
    Clusters are always exact; the number of runs (of 200) with a row-count difference or a boundary more than one mel
    frame off must not exceed the committed measurement (profiles/README.md, parity table) plus box-to-box slack."""
    from tools.parity_sweep import score
    from whisperseg_amd.model import WhisperSegmenter
    res = score(WhisperSegmenter(MODEL_DIR, device="cuda", device_ids=[0], dtype=dtype), sweep)
    print(dtype, json.dumps({k: v for k, v in res.items() if not k.endswith("_runs") or isinstance(v, int)}))
    bad = res["structure_mismatch_runs"] + res["beyond_one_frame_runs"]
    assert len(bad) <= MAX_BAD_RUNS[dtype], bad[:5]
    assert res["cluster_mismatch_rows"] == 0


# ---- the HELD-OUT sweeps (r06) ------------------------------------------------------------------------------------
# sweep 2: 1 000 recordings (250 new seeds x trials {1, 3} x beams {1, 4}) of a second, independently trained fixture model of another
# shape with full-mantissa fp32 weights (tests/golden/tiny_model2, tools/tiny_model.py variant "tiny2"), rows recorded from the reference
# by tools/make_golden.py --only sweep2.  No precision format was chosen on it: they were frozen before it was recorded.  It put f16m6,
# the default of r04-r05, outside the tolerance on 2 of its recordings; the pre-registered fallback (f16m6 with 24-bit cross K / V rows,
# then f16x3) was evaluated on it, and sweep 3 (--only sweep3: 1 000 further recordings, seeds 7000..7249) is the fresh test of the
# outcome: f16m6 with 24-bit rows failed it too (2 recordings; profiles/r06_fallback_f16m6_k24.json), f16x3 / bf16x3 reproduce every row.
# name -> (rows, fixture model, signal family).  sweep 4: 1 000 more of the second model (seeds 9000..9249); sweep 5: 1 000 recordings of
# a THIRD model (d 128, 4 + 4 layers, fp32 weights: tests/golden/tiny_model3) — the same question on another architecture; sweep 6: 1 000
# more of the third model (seeds 13000..13249), recorded while the x3 modes' cross K / V rows moved from 24-bit floats to 24-bit block
# floating point (the remedy for f16x3's one miss on sweep 5) and first scored with that format frozen; sweep 7: 1 000 more (seeds
# 15000..15249), recorded after sweep 6 had been scored.
HELDOUT = {"sweep2": ("tiny2_sweep.json", "tiny_model2", "tiny2"), "sweep3": ("tiny2_sweep3.json", "tiny_model2", "tiny2"),
           "sweep4": ("tiny2_sweep4.json", "tiny_model2", "tiny2"), "sweep5": ("tiny3_sweep.json", "tiny_model3", "tiny3"),
           "sweep6": ("tiny3_sweep6.json", "tiny_model3", "tiny3"), "sweep7": ("tiny3_sweep7.json", "tiny_model3", "tiny3")}


@pytest.fixture(scope="module", params=sorted(HELDOUT))
def heldout(request):
    rows, model, variant = HELDOUT[request.param]
    with open(os.path.join(GOLDEN, rows)) as f:
        return request.param, json.load(f), os.path.join(GOLDEN, model), variant


def test_heldout_sweeps_are_big_enough(heldout):
    name, sweep, _, _ = heldout
    assert len(sweep) == 1000
    assert sum(len(r["expected"]["onset"]) for r in sweep) >= 5000
    assert len([r for r in sweep if r["kwargs"]["num_trials"] == 3 and r["expected"]["onset"]]) >= 400


@pytest.mark.parametrize("dtype", ["f32", "f16x3", "bf16x3"])
def test_heldout_exact_and_split_precision_modes_reproduce_every_row(gpu_lib, heldout, dtype):
    """f32: exact by construction, on every sweep.  f16x3 (the default): every row of the held-out sweeps 2-7 identical to the
    reference's — asserted as the north-star tolerance (clusters exact, boundaries within +-1 frame on EVERY recording) AND as
    bit-identical rows.  bf16x3: the same on sweeps 2-4, 6 and 7; on sweep 5 (the third model) it misses 2 recordings (see below).
    Sweep 2: every recording through its own segment() call, as the rows were recorded (~50 s per mode).  Sweeps 3-7: the POOLED path —
    two segment_batch() calls, ~1 900 windows sharing the engine's slots, admitted in whatever groups the scheduler forms — the rows
    must be the reference's either way.  profiles/r06_parity_sweeps.json holds the per-file record of all seven sweeps in every mode."""
    from tools.parity_sweep import score, score_pooled
    from whisperseg_amd.model import WhisperSegmenter
    name, sweep, model_dir, variant = heldout
    seg = WhisperSegmenter(model_dir, device="cuda", device_ids=[0], dtype=dtype)
    res = score(seg, sweep, variant) if name == "sweep2" else score_pooled(seg, sweep, variant)
    print(name, dtype, json.dumps({k: v for k, v in res.items() if not k.endswith("_runs") or isinstance(v, int)}))
    bad = res["structure_mismatch_runs"] + res["beyond_one_frame_runs"]
    if name == "sweep5" and dtype == "bf16x3":
        # The third model.  MEASURED (r06, per file and pooled alike): bf16x3 misses recording 466 (seed 11116, greedy, 3 trials: one
        # time token of one window, top-1 / top-2 margin 2.8e-5 in the fp32 oracle) and ONE more — 577 (seed 11144) with the 24-bit
        # float cross K / V rows of mid r06, 957 (seed 11239) with the 24-bit block-floating-point rows it ships with: its GEMM error
        # (bfloat16 pairs, 1.3e-5 of the logit scale) is the cause, the rows only move which near-tie falls.  This is a
        # characterisation set from that measurement, not a tolerance: the claim for bf16x3 is "6 198 of 6 200"; f16x3 and f32
        # (asserted below on every sweep) reproduce everything.
        assert len(bad) <= 2, bad[:5]
        assert {b["seed"] for b in bad} <= {11116, 11239}, bad
        assert res["exact_runs"] >= len(sweep) - 3
        return
    assert bad == [], bad[:3]
    assert res["cluster_mismatch_rows"] == 0 and res["within_tolerance_runs"] == len(sweep)
    assert res["exact_runs"] == len(sweep)


@pytest.mark.parametrize("variant,model,first_seed", [("tiny2", "tiny_model2", 5000), ("tiny3", "tiny_model3", 11000)])
def test_default_mode_logit_error_on_the_trained_models(gpu_lib, variant, model, first_seed):
    """The precision of the default mode where it matters — on TRAINED models, whose cross-attention puts its probability on single
    encoder positions: first-step logits of f16x3 against the exact f32 mode over the 84 windows of 24 sweep recordings.  MEASURED
    (profiles/r06_trained_logit_error.txt; max / mean |diff| on a scale of 17 / 21): 1.5e-5 / 6.7e-7 and 1.1e-5 / 8.2e-7 with the shipped
    24-bit block-floating-point cross K / V rows — the same as with fp32 rows; the 24-bit FLOAT rows of r03 - mid r06 gave 4.8e-5 / 2.2e-6
    and 2.9e-5 / 2.5e-6 (and cost one of 4 200 sweep recordings), 16-bit block floating point 1.5e-4 / 7.1e-6.  The bounds sit between the
    shipped format and the next worse one, so a coarser row format (or a GEMM that loses bits) fails here before it flips a token."""
    from tools.trained_logit_error import measure
    r = measure(variant, model, first_seed, 24, ["f16x3"])["f16x3"]
    print(json.dumps(r))
    assert r["windows"] == 84 and r["argmax_equal"] == 84
    assert r["max_abs_logit_err"] <= 2.4e-5 and r["mean_abs_logit_err"] <= 1.3e-6, r


def test_default_mode_is_the_split_mode_that_reproduces_every_sweep():
    from whisperseg_amd.model import DEFAULT_DTYPE
    assert DEFAULT_DTYPE == "f16x3"


@pytest.mark.parametrize("dtype", ["f16m6", "f16", "bf16"])
def test_heldout_faster_modes_stay_inside_their_measured_envelope(gpu_lib, heldout, dtype):
    """CHARACTERISATION of the modes that are faster and outside the tolerance: recordings (of 1 000) with a row-count / cluster
    difference or a boundary more than one mel frame off must not exceed the committed per-file measurement (f16m6 2 / 3 / 1 / 3 / 0 / 1, f16 14 /
    15 / 26 / 28 / 12 / 17, bf16 95 / 91 / 91 / 104 / 103 / 81 on sweeps 2 / 3 / 4 / 5 / 6 / 7, profiles/r06_parity_sweeps.json) plus slack for the pooled path used here (other neighbours, other
    near-tie resolutions) and box-to-box differences."""
    from tools.parity_sweep import score_pooled
    from whisperseg_amd.model import WhisperSegmenter
    name, sweep, model_dir, variant = heldout
    res = score_pooled(WhisperSegmenter(model_dir, device="cuda", device_ids=[0], dtype=dtype), sweep, variant)
    print(name, dtype, json.dumps({k: v for k, v in res.items() if not k.endswith("_runs") or isinstance(v, int)}))
    bad = res["structure_mismatch_runs"] + res["beyond_one_frame_runs"]
    assert len(bad) <= {"f16m6": 8, "f16": 45, "bf16": 140}[dtype], bad[:5]


# runs (of 200) allowed outside "clusters exact, boundaries within +-1 frame"; set from the measured sweeps
MAX_BAD_RUNS = {"f16": 12, "bf16": 36}      # measured: f16 10 / bf16 28 (r02), f16 8 / bf16 30 (r03: another log-mel kernel moves other near-ties)

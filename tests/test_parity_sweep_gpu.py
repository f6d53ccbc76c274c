"""North-star parity over 200 recordings (tests/golden/tiny_sweep.json, rows recorded from the reference's own segment()
on HF fp32): the exact mode f32 and the split-precision modes — f16m6 (the product default), f16x3, bf16x3 — must reproduce EVERY
row (0 recordings beyond +-1 mel frame, clusters bit-exact); the plain 16-bit modes f16 / bf16 are outside the north-star tolerance
and are characterised: they must stay inside the measured envelope committed in profiles/ (scored with tools/parity_sweep.py)."""
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
    """The fast parity modes: GEMM operands as hi + lo 16-bit pairs, three MFMAs per product (f16x3 / bf16x3) or hi*hi on the
    half matrix cores + both cross terms on the fp6 MX matrix cores (f16m6), fp32 everywhere else.  North star: clusters exact, boundaries within +-1 mel frame — on EVERY recording of the sweep."""
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


# ---- the HELD-OUT sweep (r06) -------------------------------------------------------------------------------------
# 1 000 recordings (250 new seeds x trials {1, 3} x beams {1, 4}) of a second, independently trained fixture model of another shape with
# full-mantissa fp32 weights (tests/golden/tiny_model2, tools/tiny_model.py variant "tiny2"), rows recorded from the reference by
# tools/make_golden.py --only sweep2.  No precision format was chosen on it: they were frozen before it was recorded.
MODEL2_DIR = os.path.join(GOLDEN, "tiny_model2")


@pytest.fixture(scope="module")
def sweep2():
    with open(os.path.join(GOLDEN, "tiny2_sweep.json")) as f:
        return json.load(f)


def test_heldout_sweep_is_big_enough(sweep2):
    assert len(sweep2) == 1000
    assert sum(len(r["expected"]["onset"]) for r in sweep2) >= 4000
    assert len([r for r in sweep2 if r["kwargs"]["num_trials"] == 3 and r["expected"]["onset"]]) >= 250


def test_heldout_f32_mode_reproduces_every_row(gpu_lib, sweep2):
    from tools.parity_sweep import score
    from whisperseg_amd.model import WhisperSegmenter
    res = score(WhisperSegmenter(MODEL2_DIR, device="cuda", device_ids=[0], dtype="f32"), sweep2, "tiny2")
    assert res["exact_runs"] == len(sweep2), (res["structure_mismatch_runs"][:3], res["beyond_one_frame_runs"][:3])


@pytest.mark.parametrize("dtype", ["f16m6", "f16x3", "bf16x3"])
def test_heldout_split_precision_modes_meet_the_north_star_tolerance(gpu_lib, sweep2, dtype):
    from tools.parity_sweep import score
    from whisperseg_amd.model import WhisperSegmenter
    res = score(WhisperSegmenter(MODEL2_DIR, device="cuda", device_ids=[0], dtype=dtype), sweep2, "tiny2")
    print(dtype, json.dumps({k: v for k, v in res.items() if not k.endswith("_runs") or isinstance(v, int)}))
    assert res["structure_mismatch_runs"] == [] and res["beyond_one_frame_runs"] == [], (res["structure_mismatch_runs"][:3], res["beyond_one_frame_runs"][:3])
    assert res["cluster_mismatch_rows"] == 0
    assert res["within_tolerance_runs"] == len(sweep2)


# runs (of 200) allowed outside "clusters exact, boundaries within +-1 frame"; set from the measured sweeps
MAX_BAD_RUNS = {"f16": 12, "bf16": 36}      # measured: f16 10 / bf16 28 (r02), f16 8 / bf16 30 (r03: another log-mel kernel moves other near-ties)

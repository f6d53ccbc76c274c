"""Pins oracle/frontend.py against golden vectors recorded from the reference (tools/make_golden.py)."""
import hashlib
import json
import os

import numpy as np
import pytest

import golden_inputs as GI
from oracle import frontend as O


@pytest.fixture(scope="module")
def fe_golden(golden_dir):
    with open(os.path.join(golden_dir, "frontend.json")) as f:
        return json.load(f)


def test_n_fft_ladder(fe_golden):
    for sr, n_fft in fe_golden["n_fft"].items():
        assert O.get_n_fft_given_sr(int(sr)) == n_fft


def test_mel_filterbank_bit_exact(fe_golden):
    for key, g in fe_golden["mel_filters"].items():
        sr, min_f = (int(v) for v in key.split("_"))
        fb = O.mel_filter_bank_slaney(sr, g["n_fft"], min_f)
        assert list(fb.shape) == g["shape"]
        assert hashlib.sha256(fb.tobytes()).hexdigest() == g["sha256"]
        for r, row in g["rows"].items():
            assert fb[int(r)].tolist() == row


@pytest.mark.parametrize("case", GI.LOGMEL_CASES, ids=[c[0] for c in GI.LOGMEL_CASES])
def test_logmel_matches_reference(golden_dir, case):
    name, sr, sts, min_f, kind, seed = case
    z = np.load(os.path.join(golden_dir, "logmel.npz"))
    x = GI.signal(kind, GI.window_len(sr, sts), sr, seed)
    got = O.logmel_window(x, sr, sts, min_f)
    assert list(got.shape) == z[name + "__shape"].tolist()
    sub = got[:, ::GI.COL_STRIDE]
    # numpy float64 path (what the pinned transformers 4.38.2 runs): identical arithmetic
    assert np.max(np.abs(sub - z[name + "__np"])) <= 1e-6
    # torch float32 path (what transformers 5.15 dispatches to): HF's own cross-path tolerance class
    assert np.max(np.abs(sub - z[name + "__default"])) <= 1e-4


def test_window_tables(golden_dir):
    with open(os.path.join(golden_dir, "windows.json")) as f:
        cases = json.load(f)
    assert len(cases) == len(GI.WINDOW_TABLE_CASES)
    for c in cases:
        rows = O.window_table(c["n"], c["sr"], c["sts"], c["trials"])
        got = [[r[0], r[4], r[5] / c["sr"]] for r in rows]
        assert got == c["table"], c


def test_known_answers():
    """SURVEY §8c probes: all-zero window -> -1.5 everywhere; 44.1 kHz -> hop 110 and 1002 raw frames;
    lengths {0, 1, 80000, 160000, 160001} at 32 kHz / 2.5 ms -> {1, 1, 1, 2, 3} windows."""
    f = O.logmel_window(np.zeros(160000, np.float32), 16000, 0.01)
    assert np.all(f == -1.5)
    assert O.logmel_window(np.zeros(110250, np.float32), 44100, 0.0025).shape == (80, 1002)
    for n, want in [(0, 1), (1, 1), (80000, 1), (160000, 2), (160001, 3)]:
        assert len(O.window_table(n, 32000, 0.0025, 1)) == want

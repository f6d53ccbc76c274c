"""The PRODUCT's host-side front-end tables (whisperseg_amd/audio_utils.py) against the vectors recorded from the
reference (G1 n_fft ladder, G2 slaney filterbanks; tools/make_golden.py) — the oracle is pinned separately in
tests/test_oracle_frontend.py.  No GPU needed: these are the host computations that feed the log-mel kernel."""
import hashlib
import json
import os

import numpy as np
import pytest


@pytest.fixture(scope="module")
def fe_golden(golden_dir):
    with open(os.path.join(golden_dir, "frontend.json")) as f:
        return json.load(f)


def test_product_n_fft_ladder(fe_golden):
    from whisperseg_amd.audio_utils import get_n_fft_given_sr
    assert len(fe_golden["n_fft"]) >= 12
    for sr, n_fft in fe_golden["n_fft"].items():
        assert get_n_fft_given_sr(int(sr)) == n_fft, sr


def test_product_mel_filterbank_bit_exact(fe_golden):
    from whisperseg_amd.audio_utils import get_n_fft_given_sr, slaney_mel_filters
    for key, g in fe_golden["mel_filters"].items():
        sr, min_f = (int(v) for v in key.split("_"))
        assert get_n_fft_given_sr(sr) == g["n_fft"]
        fb = np.asarray(slaney_mel_filters(sr, g["n_fft"], min_f, sr // 2), dtype=np.float64)
        assert list(fb.shape) == g["shape"]
        assert hashlib.sha256(np.ascontiguousarray(fb).tobytes()).hexdigest() == g["sha256"], key
        for r, row in g["rows"].items():
            assert fb[int(r)].tolist() == row
        assert np.array_equal(fb.sum(0), np.asarray(g["col_sums"]))

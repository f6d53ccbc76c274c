"""Known-answer harness for REAL checkpoints.

No WhisperSeg checkpoint exists offline, so every other parity test is pinned on a tiny model trained here.  The day a
checkpoint directory is supplied this test pins the product on the reference's PUBLISHED answer: reference README.md:314-324
prints the 17 rows that `segmenter.segment(audio, sr=32000, spec_time_step=0.0025)` gives for
data/example_subset/Zebra_finch/test_adults/zebra_finch_g17y2U-f00007.wav with the `whisperseg-large-ms` checkpoint
(tests/golden/known_answer_zebra_finch.json holds those rows; the wav — 7.2 s, natively 32 kHz, so `librosa.load(sr=32000)` is a
plain decode and no resampler is involved — is committed beside it as data).

    WSEG_CHECKPOINT_DIR=/path/to/whisperseg-large-ms python -m pytest tests/test_known_answer_gpu.py -m gpu

Skipped when $WSEG_CHECKPOINT_DIR is unset.  Bar = the north star: the same number of rows, cluster labels exact, every
boundary within +-1 mel frame (spec_time_step seconds) — in the segmenter's DEFAULT mode and in the exact-parity f32 mode
($WSEG_KNOWN_ANSWER_MODES overrides the list, e.g. "f16,bf16" to characterise the plain modes on real weights)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

CKPT = os.environ.get("WSEG_CHECKPOINT_DIR")
MODES = [m for m in os.environ.get("WSEG_KNOWN_ANSWER_MODES", "default,f32").split(",") if m]


def load_case():
    with open(os.path.join(GOLDEN, "known_answer_zebra_finch.json")) as f:
        return json.load(f)


def test_fixture_is_well_formed():
    """Runs on CPU (no checkpoint, no device): the published rows and the recording they refer to."""
    from whisperseg_amd.wavio import load_wav
    case = load_case()
    exp = case["expected"]
    assert len(exp["onset"]) == len(exp["offset"]) == len(exp["cluster"]) == 17
    assert all(b > a for a, b in zip(exp["onset"], exp["offset"])) and exp["onset"] == sorted(exp["onset"])
    audio, sr = load_wav(os.path.join(GOLDEN, case["wav"]))
    assert sr == case["sr"] == 32000 and audio.dtype == np.float32 and exp["offset"][-1] < len(audio) / sr < 8.0


@pytest.mark.gpu
@pytest.mark.skipif(not CKPT, reason="set WSEG_CHECKPOINT_DIR to a WhisperSeg checkpoint directory (e.g. whisperseg-large-ms)")
@pytest.mark.parametrize("mode", MODES)
def test_readme_rows_of_the_published_checkpoint(gpu_lib, mode):
    from whisperseg_amd.model import WhisperSegmenter
    from whisperseg_amd.wavio import load_wav
    case = load_case()
    audio, sr = load_wav(os.path.join(GOLDEN, case["wav"]))
    seg = WhisperSegmenter(CKPT, device="cuda", device_ids=[0], dtype=None if mode == "default" else mode)
    got = seg.segment(audio, sr=case["sr"], spec_time_step=case["spec_time_step"])
    want = case["expected"]
    assert len(got["onset"]) == len(want["onset"]), (mode, got)
    assert [str(c) for c in got["cluster"]] == want["cluster"], mode
    dev = np.abs(np.array(got["onset"] + got["offset"]) - np.array(want["onset"] + want["offset"]))
    # README prints 3 decimals (precision_bits = 3): half a milli-second of print rounding on top of one mel frame
    assert dev.max() <= case["spec_time_step"] + 5e-4 + 1e-9, (mode, float(dev.max()), got)

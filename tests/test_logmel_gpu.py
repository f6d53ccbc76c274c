"""HIP log-mel front-end (through the C-ABI) vs the oracle, same seeded inputs.

Tolerance: 1e-4 absolute on the normalised features.  The reference pipeline's own two HF paths
(numpy float64 vs torch float32) differ by ~2e-6; one mel frame in time is a column, so a 1e-4 value
error is far inside the +-1-frame boundary tolerance of the north star."""
import numpy as np
import pytest
import torch

from oracle import frontend as O

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _signal(kind, n, sr, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / sr
    if kind == "sine_noise":
        return (0.1 * np.sin(2 * np.pi * 440 * t) + 0.01 * rng.standard_normal(n)).astype(np.float32)
    if kind == "chirp":
        return (0.2 * np.sin(2 * np.pi * (200 + 0.4 * sr / 2 * t / max(t[-1], 1e-9)) * t)).astype(np.float32)
    if kind == "silence":
        return np.zeros(n, np.float32)
    if kind == "impulse":
        x = np.zeros(n, np.float32)
        x[n // 3] = 1.0
        return x
    raise ValueError(kind)


CASES = [  # (sr, sts, min_f, kind)
    (16000, 0.01, 0, "sine_noise"), (16000, 0.03, 0, "chirp"), (32000, 0.0025, 0, "sine_noise"),
    (48000, 0.0025, 0, "chirp"), (44100, 0.0025, 0, "sine_noise"), (300000, 0.0005, 35000, "sine_noise"),
    (16000, 0.01, 0, "silence"), (32000, 0.0025, 0, "impulse"), (96000, 0.001, 500, "sine_noise"),
    (384000, 0.0005, 0, "chirp"),
]


@pytest.mark.parametrize("sr,sts,min_f,kind", CASES)
def test_single_window_matches_oracle(gpu_lib, sr, sts, min_f, kind):
    from whisperseg_amd.audio_utils import WhisperSegFeatureExtractor
    L = int(1000 * sts * sr)
    x = _signal(kind, L, sr, 0)
    want = O.logmel_window(x, sr, sts, min_f)[:, :1000]
    ext = WhisperSegFeatureExtractor(sr, sts, min_frequency=min_f, device="cuda:0")
    got = ext.extract_windows(torch.from_numpy(x).cuda(), torch.zeros(1, dtype=torch.int64), L)[0].cpu().numpy()
    assert got.shape == (80, 1000)
    n = min(want.shape[1], 1000)
    assert np.max(np.abs(got[:, :n] - want[:, :n])) <= TOL
    if kind == "silence":
        assert np.all(got == -1.5)          # (log10(1e-10) + 4) / 4, SURVEY §8c known answer


@pytest.mark.parametrize("n_audio,num_trials", [(0, 1), (1, 1), (80000, 1), (160000, 3), (160001, 3), (200123, 2)])
def test_recording_windows_match_oracle(gpu_lib, n_audio, num_trials):
    """All windows of a recording (multi-trial left padding, ragged tail) in one call."""
    from whisperseg_amd.audio_utils import WhisperSegFeatureExtractor
    sr, sts = 32000, 0.0025
    x = _signal("sine_noise", max(n_audio, 1), sr, 3)[:n_audio]
    want = O.sliced_audio_features(x, sr, 0, sts, num_trials)
    table = O.window_table(n_audio, sr, sts, num_trials)
    L = int(1000 * sts * sr)
    starts = torch.tensor([pos - n_pad for (_, pos, n_pad, _, _, _) in table], dtype=torch.int64)
    ext = WhisperSegFeatureExtractor(sr, sts, min_frequency=0, device="cuda:0")
    got = ext.extract_windows(torch.from_numpy(x).cuda(), starts, L).cpu().numpy()
    assert got.shape[0] == len(want)
    for g, (_, _, f, _) in zip(got, want):
        assert np.max(np.abs(g - f)) <= TOL


def test_hf_style_call(gpu_lib):
    from whisperseg_amd.audio_utils import WhisperSegFeatureExtractor
    sr, sts = 44100, 0.0025
    L = int(1000 * sts * sr)
    x = _signal("sine_noise", L, sr, 5)
    ext = WhisperSegFeatureExtractor(sr, sts, device="cuda:0")
    f = ext(x, sampling_rate=sr, padding="do_not_pad")["input_features"][0]
    want = O.logmel_window(x, sr, sts)
    assert f.shape == want.shape == (80, 1002)      # hop 110 -> 1002 frames before truncation
    assert np.max(np.abs(f - want)) <= TOL

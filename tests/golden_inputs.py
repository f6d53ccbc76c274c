"""Deterministic synthetic inputs shared by tools/make_golden.py (which records the reference's outputs
for them) and the tests (which regenerate the same inputs and compare).  numpy's PCG64 streams are
bit-reproducible across machines."""
import numpy as np

LOGMEL_CASES = [  # name, sr, spec_time_step, min_frequency, kind, seed
    ("sine16k_10ms", 16000, 0.01, 0, "sine_noise", 0),
    ("chirp16k_30ms", 16000, 0.03, 0, "chirp", 1),
    ("sine32k", 32000, 0.0025, 0, "sine_noise", 2),
    ("chirp48k", 48000, 0.0025, 0, "chirp", 3),
    ("sine44k1", 44100, 0.0025, 0, "sine_noise", 4),
    ("mouse300k", 300000, 0.0005, 35000, "sine_noise", 5),
    ("silence16k", 16000, 0.01, 0, "silence", 6),
    ("impulse32k", 32000, 0.0025, 0, "impulse", 7),
]
COL_STRIDE = 4          # golden log-mel windows store every 4th column


def signal(kind, n, sr, seed):
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


def window_len(sr, sts, cols=1000):
    return int(cols * sts * sr)


WINDOW_TABLE_CASES = [  # sr, sts, n_samples, num_trials
    (32000, 0.0025, 0, 1), (32000, 0.0025, 1, 1), (32000, 0.0025, 80000, 1), (32000, 0.0025, 160000, 1),
    (32000, 0.0025, 160001, 1), (32000, 0.0025, 0, 3), (32000, 0.0025, 1, 3), (32000, 0.0025, 79999, 3),
    (32000, 0.0025, 160000, 3), (32000, 0.0025, 160001, 3), (16000, 0.01, 1234567, 3), (44100, 0.0025, 300000, 3),
    (48000, 0.0025, 500000, 2), (300000, 0.0005, 700001, 3), (16000, 0.03, 480000 * 3 + 5, 5), (16000, 0.001, 40000, 3),
]


def tiny_recording(seed, n_windows=3, tail=0.37, variant="tiny"):
    """Concatenated synthetic tone-burst clips (tools/tiny_model.synth_clip) -> a multi-window recording.
    variant: the signal family of the fixture model the recording is for (tools/tiny_model.VARIANTS)."""
    from tools import tiny_model as TM
    rng = np.random.default_rng(seed)
    parts = [TM.synth_clip(rng, variant=variant)[0] for _ in range(n_windows)]
    x = np.concatenate(parts)
    cut = int(len(parts[-1]) * (1.0 - tail))
    return x[: len(x) - cut].astype(np.float32)

"""ORACLE (test infrastructure, NOT product code) — plain-numpy restatement of rational polyphase resampling.

The reference's resampler is `librosa.load(path, sr=target)` (scripts/segment.py:48,61; evaluate.py:58): a third-party,
un-pinned dependency (librosa -> soxr/resampy) that is absent from the image, so PARITY WITH THE REFERENCE IS UNPINNED for
this row.  The restatement below follows the published polyphase algorithm as implemented by
`scipy.signal.resample_poly(x, up, down, window=('kaiser', 5.0))` (scipy is in the image) and
tests/test_resample.py pins it against scipy on seeded inputs."""
import math

import numpy as np


def resample_poly_ref(x, sr_in, sr_out):
    x = np.asarray(x, dtype=np.float32)
    g = math.gcd(int(sr_in), int(sr_out))
    up, down = int(sr_out) // g, int(sr_in) // g
    if up == down == 1:
        return x.copy()
    n_in = len(x)
    n_out = -(-n_in * up // down)
    max_rate = max(up, down)
    half_len = 10 * max_rate
    n = 2 * half_len + 1
    m = np.arange(n, dtype=np.float64) - half_len
    h = (1.0 / max_rate) * np.sinc(m / max_rate) * np.kaiser(n, 5.0)
    h /= h.sum()
    h = h.astype(np.float32) * np.float32(up)
    pre_pad = down - half_len % down
    pre_remove = (half_len + pre_pad) // down
    y = np.zeros(n_out, dtype=np.float64)
    hd, xd = h.astype(np.float64), x.astype(np.float64)
    for i in range(n_out):
        c = (i + pre_remove) * down - pre_pad
        k_hi = min(c // up, n_in - 1)
        k_lo = max(0, -(-(c - n + 1) // up))
        if k_hi >= k_lo:
            k = np.arange(k_lo, k_hi + 1)
            y[i] = np.dot(hd[c - k * up], xd[k])
    return y.astype(np.float32)

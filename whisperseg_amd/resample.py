"""Sample-rate conversion on the GPU (audio ingest).

The reference resamples implicitly through `librosa.load(path, sr=target)` (scripts/segment.py:48,61, evaluate.py:58):
a third-party resampler that is neither pinned nor installed here.  This module implements the standard rational
polyphase scheme — Kaiser(beta=5)-windowed sinc low-pass of half-length 10*max(up, down), unity DC gain times `up`,
centred output of ceil(n*up/down) samples — which is also what `scipy.signal.resample_poly` computes; the arithmetic runs
in libwseg (`wseg_resample_f32`)."""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib


def design_taps(up, down, beta=5.0):
    """float32 FIR taps (already scaled by `up`) + half length, for the reduced ratio up/down."""
    max_rate = max(up, down)
    half_len = 10 * max_rate
    n = 2 * half_len + 1
    m = np.arange(n, dtype=np.float64) - half_len
    cutoff = 1.0 / max_rate
    h = cutoff * np.sinc(cutoff * m) * np.kaiser(n, beta)
    h /= h.sum()
    h = h.astype(np.float32)
    h *= np.float32(up)
    return h, half_len


def plan(n_in, sr_in, sr_out):
    g = math.gcd(int(sr_in), int(sr_out))
    up, down = int(sr_out) // g, int(sr_in) // g
    n_out = -(-n_in * up // down)
    taps, half_len = design_taps(up, down)
    pre_pad = down - half_len % down
    pre_remove = (half_len + pre_pad) // down
    return dict(up=up, down=down, n_out=n_out, taps=taps, pre_pad=pre_pad, pre_remove=pre_remove)


_TAPS = {}


def resample(audio, sr_in, sr_out, device="cuda"):
    """audio: float32 numpy array or device tensor [N] at sr_in -> float32 device tensor at sr_out."""
    x = audio if torch.is_tensor(audio) else torch.as_tensor(np.ascontiguousarray(audio, dtype=np.float32))
    x = x.to(device=device, dtype=torch.float32).contiguous()
    if int(sr_in) == int(sr_out):
        return x.clone()
    lib = _lib.load(require_device=True)
    p = plan(int(x.numel()), sr_in, sr_out)
    key = (p["up"], p["down"], str(x.device))
    if key not in _TAPS:
        _TAPS[key] = torch.from_numpy(p["taps"]).to(x.device)
    taps = _TAPS[key]
    y = torch.empty(p["n_out"], dtype=torch.float32, device=x.device)
    if p["n_out"] and x.numel():
        with torch.cuda.device(x.device):
            _lib.check(lib.wseg_resample_f32(x.data_ptr(), int(x.numel()), taps.data_ptr(), int(taps.numel()), p["up"], p["down"],
                                             p["pre_pad"], p["pre_remove"], y.data_ptr(), p["n_out"], _lib.stream_ptr()))
    else:
        y.zero_()
    return y

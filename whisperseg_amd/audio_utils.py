"""Audio front-end: host-side mirror of reference audio_utils.py:32-76 over the HIP log-mel kernels.

`WhisperSegFeatureExtractor` keeps the reference constructor signature and the HF-style call
(`extractor(clip, sampling_rate=sr, padding="do_not_pad")["input_features"][0]`), but the STFT / mel /
log / normalise arithmetic runs in libwseg (`wseg_logmel_f32`).  The batched entry point
`extract_windows` computes every window of a recording in one launch pair, with window slicing,
zero padding and the 1000-column truncation folded into the kernel (reference model.py:138-161).
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib

N_MELS = 80


def get_n_fft_given_sr(sr):
    """FFT-size ladder by sampling rate (reference audio_utils.py:32-43)."""
    for limit, n_fft in ((32000, 512), (80000, 1024), (150000, 2048), (300000, 4096)):
        if sr <= limit:
            return n_fft
    return 8192


def _slaney_hz_to_mel(f):
    f = np.asarray(f, dtype=np.float64)
    lin = 3.0 * f / 200.0
    with np.errstate(divide="ignore", invalid="ignore"):
        log = 15.0 + np.log(np.maximum(f, 1e-300) / 1000.0) * (27.0 / np.log(6.4))
    return np.where(f >= 1000.0, log, lin)


def _slaney_mel_to_hz(m):
    m = np.asarray(m, dtype=np.float64)
    lin = 200.0 * m / 3.0
    log = 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0))
    return np.where(m >= 15.0, log, lin)


def slaney_mel_filters(sr, n_fft, min_frequency, max_frequency, n_mels=N_MELS):
    """float64 [n_fft/2+1, n_mels] slaney-scale, slaney-normalised triangles on [min_f, max_f] with
    FFT bin centres linspace(0, sr//2, n_bins)  (reference audio_utils.py:68-76 -> HF mel_filter_bank)."""
    n_bins = n_fft // 2 + 1
    edges_mel = np.linspace(float(_slaney_hz_to_mel(min_frequency)), float(_slaney_hz_to_mel(max_frequency)), n_mels + 2)
    edges = _slaney_mel_to_hz(edges_mel)
    bins = np.linspace(0, sr // 2, n_bins)
    width = np.diff(edges)
    rel = edges[None, :] - bins[:, None]
    falling = -rel[:, :-2] / width[:-1]
    rising = rel[:, 2:] / width[1:]
    fb = np.maximum(0.0, np.minimum(falling, rising))
    fb *= (2.0 / (edges[2:] - edges[:-2]))[None, :]
    return fb


class _DeviceTables:
    """Device-resident tables for one (n_fft, hop, filterbank) configuration."""

    def __init__(self, n_fft, hop, mel_filters, n_cols, device):
        self.n_fft, self.hop, self.n_cols = n_fft, hop, n_cols
        win = np.hanning(n_fft + 1)[:-1]
        k = np.arange(n_fft // 2, dtype=np.float64)
        ang = 2.0 * np.pi * k / n_fft
        tw = np.stack([np.cos(ang), -np.sin(ang)], axis=1)
        starts, counts, offsets, weights = [], [], [], []
        for m in range(mel_filters.shape[1]):
            nz = np.nonzero(mel_filters[:, m])[0]
            if len(nz) == 0:
                s, c = 0, 0
            else:
                s, c = int(nz[0]), int(nz[-1] - nz[0] + 1)
            starts.append(s)
            counts.append(c)
            offsets.append(len(weights))
            weights.extend(mel_filters[s:s + c, m].tolist())
        if not weights:
            weights = [0.0]
        dev = torch.device(device)
        self.window = torch.tensor(win, dtype=torch.float32, device=dev)
        self.twiddle = torch.tensor(tw, dtype=torch.float32, device=dev).contiguous()
        self.mel_start = torch.tensor(starts, dtype=torch.int32, device=dev)
        self.mel_count = torch.tensor(counts, dtype=torch.int32, device=dev)
        self.mel_offset = torch.tensor(offsets, dtype=torch.int32, device=dev)
        self.mel_weight = torch.tensor(weights, dtype=torch.float32, device=dev)
        self.desc = _lib.LogmelDesc(
            n_fft=n_fft, hop=hop, n_mels=mel_filters.shape[1], n_cols=n_cols,
            window=self.window.data_ptr(), twiddle=self.twiddle.data_ptr(),
            mel_start=self.mel_start.data_ptr(), mel_count=self.mel_count.data_ptr(),
            mel_offset=self.mel_offset.data_ptr(), mel_weight=self.mel_weight.data_ptr())


class WhisperSegFeatureExtractor:
    """Same constructor as reference audio_utils.py:45-76; the arithmetic runs on the GPU."""

    def __init__(self, sr, spec_time_step, min_frequency=None, max_frequency=None, chunk_length=30,
                 n_cols=1000, device=None):
        self.sampling_rate = sr
        self.hop_length = int(spec_time_step * sr)
        self.n_fft = get_n_fft_given_sr(sr)
        self.chunk_length = chunk_length
        self.n_samples = chunk_length * sr
        self.feature_size = N_MELS
        self.n_cols = n_cols
        if min_frequency is None:
            min_frequency = 0
        if max_frequency is None:
            max_frequency = sr // 2
        if self.hop_length <= 0:
            raise ValueError("spec_time_step * sr must be at least 1 sample")
        self.mel_filters = slaney_mel_filters(sr, self.n_fft, min_frequency, max_frequency)
        self._device = device
        self._tables = None

    def _get_tables(self, device):
        if self._tables is None or self._tables.window.device != torch.device(device):
            self._tables = _DeviceTables(self.n_fft, self.hop_length, self.mel_filters, self.n_cols, device)
        return self._tables

    def extract_windows(self, audio, win_start, win_len):
        """audio: float32 device tensor [N]; win_start: int64 device tensor [W] (may be negative / past
        the end: zero-filled); win_len: samples per window.  Returns float32 device tensor [W, 80, n_cols]."""
        lib = _lib.load(require_device=True)
        if not audio.is_cuda:
            raise _lib.WsegError("extract_windows needs device-resident audio (no CPU path)")
        audio = audio.contiguous().to(torch.float32)
        win_start = win_start.to(device=audio.device, dtype=torch.int64).contiguous()
        W = int(win_start.numel())
        t = self._get_tables(audio.device)
        out = torch.empty((W, N_MELS, self.n_cols), dtype=torch.float32, device=audio.device)
        if W == 0:
            return out
        nbytes = lib.wseg_logmel_scratch_bytes(C.byref(t.desc), W, int(win_len))
        scratch = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=audio.device)
        with torch.cuda.device(audio.device):
            _lib.check(lib.wseg_logmel_f32(C.byref(t.desc), audio.data_ptr() if audio.numel() else None,
                                           int(audio.numel()), win_start.data_ptr(), W, int(win_len),
                                           scratch.data_ptr(), nbytes, out.data_ptr(), _lib.stream_ptr()))
        return out

    def __call__(self, raw_speech, sampling_rate=None, padding="do_not_pad", **kwargs):
        """HF-style single-clip call used by reference model.py:152: returns all floor(L/hop) frames
        (no 1000-column truncation), as {"input_features": [float32 [80, frames]]}."""
        if sampling_rate is not None and sampling_rate != self.sampling_rate:
            raise ValueError(f"extractor built for sr={self.sampling_rate}, got {sampling_rate}")
        if padding != "do_not_pad":
            raise ValueError("only padding='do_not_pad' is supported (as the reference calls it)")
        dev = self._device or "cuda"
        x = torch.as_tensor(np.asarray(raw_speech, dtype=np.float32)).to(dev)
        L = int(x.numel())
        n_frames = L // self.hop_length
        saved = self.n_cols, self._tables
        try:
            self.n_cols, self._tables = max(n_frames, 1), None
            feats = self.extract_windows(x, torch.zeros(1, dtype=torch.int64, device=dev), L)
        finally:
            self.n_cols, self._tables = saved
        return {"input_features": [feats[0, :, :n_frames].cpu().numpy()]}


_EXTRACTOR_CACHE = {}


def get_feature_extractor(sr, spec_time_step, min_frequency, chunk_length, n_cols, device):
    """The reference rebuilds its extractor on every segment() call (model.py:128); the filterbank and
    device tables only depend on this key, so they are cached."""
    key = (sr, float(spec_time_step), min_frequency, chunk_length, n_cols, str(device))
    ext = _EXTRACTOR_CACHE.get(key)
    if ext is None:
        ext = WhisperSegFeatureExtractor(sr, spec_time_step, min_frequency=min_frequency,
                                         chunk_length=chunk_length, n_cols=n_cols, device=device)
        _EXTRACTOR_CACHE[key] = ext
    return ext

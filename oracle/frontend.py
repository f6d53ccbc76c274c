"""ORACLE (test infrastructure, NOT product code) — CPU restatement of the log-mel front-end.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product path (whisperseg_amd/) never does; it fails loudly when the HIP
library is missing.

What is restated, and from where (reference = /root/reference, HF = the third-party
`transformers` wheel the reference pins at 4.38.2, requirements.txt:1):

  get_n_fft_given_sr           reference audio_utils.py:32-43
  mel_filter_bank_slaney       HF audio_utils.py:448-560 (hertz_to_mel / mel_to_hertz, slaney),
                               :540-560 (_create_triangular_filter_bank), :638-729 (mel_filter_bank)
                               as configured by reference audio_utils.py:45-76
  logmel_window                HF models/whisper/feature_extraction_whisper.py:105-133
                               (_np_extract_fbank_features) -> HF audio_utils.py:809-1017 (spectrogram),
                               :745-806 (window_function)
  sliced_audio_features        reference model.py:127-166 (get_sliced_audios_features)

Pinning: tests/test_oracle_frontend.py checks every function here against
tests/golden/frontend_*.npz|json, which tools/make_golden.py produced by importing the
reference itself (and HF) in the build container.
"""
import numpy as np

N_MELS = 80


def get_n_fft_given_sr(sr):
    """reference audio_utils.py:32-43"""
    if sr <= 32000:
        return 512
    if sr <= 80000:
        return 1024
    if sr <= 150000:
        return 2048
    if sr <= 300000:
        return 4096
    return 8192


def _hz_to_mel_slaney(freq):
    """HF audio_utils.py:448-481 with mel_scale='slaney' (scalar or array)."""
    min_log_hertz, min_log_mel = 1000.0, 15.0
    logstep = 27.0 / np.log(6.4)
    if isinstance(freq, np.ndarray):
        mels = 3.0 * freq / 200.0
        reg = freq >= min_log_hertz
        mels[reg] = min_log_mel + np.log(freq[reg] / min_log_hertz) * logstep
        return mels
    mels = 3.0 * freq / 200.0
    if freq >= min_log_hertz:
        mels = min_log_mel + np.log(freq / min_log_hertz) * logstep
    return mels


def _mel_to_hz_slaney(mels):
    """HF audio_utils.py:484-517 with mel_scale='slaney' (array)."""
    min_log_hertz, min_log_mel = 1000.0, 15.0
    logstep = np.log(6.4) / 27.0
    freq = 200.0 * mels / 3.0
    reg = mels >= min_log_mel
    freq[reg] = min_log_hertz * np.exp(logstep * (mels[reg] - min_log_mel))
    return freq


def mel_filter_bank_slaney(sr, n_fft, min_frequency=None, max_frequency=None, n_mels=N_MELS):
    """float64 [n_fft/2+1, n_mels]; reference audio_utils.py:54-76 -> HF mel_filter_bank."""
    if min_frequency is None:
        min_frequency = 0
    if max_frequency is None:
        max_frequency = sr // 2
    n_bins = 1 + n_fft // 2
    mel_min = _hz_to_mel_slaney(min_frequency)
    mel_max = _hz_to_mel_slaney(max_frequency)
    mel_freqs = np.linspace(mel_min, mel_max, n_mels + 2)
    filter_freqs = _mel_to_hz_slaney(mel_freqs)
    fft_freqs = np.linspace(0, sr // 2, n_bins)
    filter_diff = np.diff(filter_freqs)
    slopes = np.expand_dims(filter_freqs, 0) - np.expand_dims(fft_freqs, 1)
    down = -slopes[:, :-2] / filter_diff[:-1]
    up = slopes[:, 2:] / filter_diff[1:]
    fb = np.maximum(np.zeros(1), np.minimum(down, up))
    enorm = 2.0 / (filter_freqs[2:n_mels + 2] - filter_freqs[:n_mels])
    fb *= np.expand_dims(enorm, 0)
    return fb


def hann_periodic(n):
    """HF window_function(n, 'hann'): np.hanning(n+1)[:-1]."""
    return np.hanning(n + 1)[:-1]


def logmel_window(audio, sr, spec_time_step, min_frequency=None, max_frequency=None,
                  complex64_spectrum=True):
    """One window -> float32 [80, n_frames-1].

    Follows HF _np_extract_fbank_features + spectrogram literally: reflect-centre pad,
    float64 frames * periodic Hann, rFFT, spectrum stored as complex64 (HF allocates
    `np.empty(..., dtype=np.complex64)`), |X|^2 in float64, mel = max(1e-10, M^T P),
    log10, drop last frame, clamp to (window max - 8), (x + 4) / 4.
    """
    hop = int(spec_time_step * sr)
    n_fft = get_n_fft_given_sr(sr)
    fb = mel_filter_bank_slaney(sr, n_fft, min_frequency, max_frequency)
    x = np.asarray(audio, dtype=np.float32)
    x = np.pad(x, [(n_fft // 2, n_fft // 2)], mode="reflect").astype(np.float64)
    win = hann_periodic(n_fft).astype(np.float64)
    n_frames = int(1 + np.floor((x.size - n_fft) / hop))
    idx = np.arange(n_fft)[None, :] + hop * np.arange(n_frames)[:, None]
    frames = x[idx] * win[None, :]
    spec = np.fft.rfft(frames, axis=1)
    if complex64_spectrum:
        spec = spec.astype(np.complex64)
    power = np.abs(spec, dtype=np.float64) ** 2.0
    mel = np.maximum(1e-10, fb.T @ power.T)
    log_spec = np.asarray(np.log10(mel), np.float32)
    log_spec = log_spec[:, :-1]
    log_spec = np.maximum(log_spec, log_spec.max() - 8.0)
    log_spec = (log_spec + 4.0) / 4.0
    return log_spec.astype(np.float32)


def window_table(n_samples, sr, spec_time_step, num_trials, total_spec_columns=1000):
    """(trial_id, pos, num_padding_samples, padding_time, offset_time, clip_len) per window.

    reference model.py:129-149,165: window bookkeeping only (no features).
    """
    clip_duration = total_spec_columns * spec_time_step
    audio_clip_length = int(clip_duration * sr)
    rows = []
    for trial_id in range(num_trials):
        padding_time = np.round(clip_duration * trial_id / num_trials / spec_time_step) * spec_time_step
        num_padding_samples = int(padding_time * sr)
        padded_len = num_padding_samples + n_samples
        for pos in range(0, max(padded_len, 1), audio_clip_length):
            offset_time = pos / sr - padding_time
            clip_len = max(0, min(padded_len, pos + audio_clip_length) - pos)
            rows.append((trial_id, pos, num_padding_samples, float(padding_time), float(offset_time), clip_len))
    return rows


def sliced_audio_features(audio, sr, min_frequency, spec_time_step, num_trials, total_spec_columns=1000):
    """reference model.py:127-166 -> list of (trial_id, offset_time, float32[80,cols], clip_seconds)."""
    audio = np.asarray(audio, dtype=np.float32)
    clip_duration = total_spec_columns * spec_time_step
    audio_clip_length = int(clip_duration * sr)
    out = []
    for (trial_id, pos, n_pad, padding_time, offset_time, clip_len) in window_table(
            len(audio), sr, spec_time_step, num_trials, total_spec_columns):
        padded = np.concatenate([np.zeros(n_pad, np.float32), audio])
        clip = padded[pos:pos + audio_clip_length]
        clip_padded = np.concatenate([clip, np.zeros(audio_clip_length - len(clip), np.float32)])
        feats = logmel_window(clip_padded, sr, spec_time_step, min_frequency)
        feats = feats[:, :total_spec_columns]
        min_val = feats.min() if feats.shape[1] > 0 else 0
        feats = np.concatenate(
            [feats, min_val * np.ones((feats.shape[0], total_spec_columns - feats.shape[1]))], axis=1
        ).astype(np.float32)
        out.append((trial_id, offset_time, feats, len(clip) / sr))
    return out

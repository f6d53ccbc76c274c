"""Window table of a recording: which fixed-length, non-overlapping clips are decoded, per trial.

Host arithmetic of reference model.py:129-149,165 restated literally (np.round is half-to-even,
int() truncates, hop/clip lengths come from float products) — changing any of it changes outputs.
No samples are touched here: the log-mel kernel reads the windows straight out of the recording in
HBM using `start` (which may be negative or run past the end; those samples are zero).
"""
from collections import namedtuple

import numpy as np

Window = namedtuple("Window", "trial_id start offset_time clip_seconds")


def clip_length_samples(total_spec_columns, spec_time_step, sr):
    """int(clip_duration * sr) with clip_duration = total_spec_columns * spec_time_step (model.py:129,133)."""
    return int(total_spec_columns * spec_time_step * sr)


def window_table(n_samples, sr, spec_time_step, num_trials, total_spec_columns):
    """List[Window] in (trial, position) order — the order reference model.py:136-165 appends in.

    start        first sample of the window relative to the un-padded recording (pos - num_padding_samples)
    offset_time  pos / sr - padding_time                               (model.py:147)
    clip_seconds len(audio_padded[pos:pos+clip_len]) / sr              (model.py:165)
    """
    clip_duration = total_spec_columns * spec_time_step
    clip_len = int(clip_duration * sr)
    if clip_len <= 0:
        raise ValueError("window length is zero samples: spec_time_step * sr too small")
    table = []
    for trial_id in range(num_trials):
        padding_time = np.round(clip_duration * trial_id / num_trials / spec_time_step) * spec_time_step
        n_pad = int(padding_time * sr)
        padded_len = n_pad + n_samples
        # "This loop must be executed once even for zero length audio" (model.py:145-146)
        for pos in range(0, max(padded_len, 1), clip_len):
            in_clip = max(0, min(padded_len, pos + clip_len) - pos)
            table.append(Window(trial_id, pos - n_pad, pos / sr - padding_time, in_clip / sr))
    return table


def shard_bounds(n_items, n_shards):
    """Contiguous split used by the reference's device fan-out: ceil(N / n_devices) items per shard,
    in order (model.py:172-175).  Returns [(lo, hi)], possibly fewer than n_shards entries."""
    if n_items <= 0:
        return []
    per = int(np.ceil(n_items / n_shards))
    return [(lo, min(n_items, lo + per)) for lo in range(0, n_items, per)]

"""Constants shared by the host side (mirror of reference utils.py:5)."""

# One decoder time token spans two spectrogram columns (reference utils.py:5).
RATIO_DECODING_TIME_STEP_TO_SPEC_TIME_STEP = 2

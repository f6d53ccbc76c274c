"""whisperseg_amd — MI355X-native hot path for WhisperSeg-style segmentation.

Host side (Python, mirrors reference model.py / audio_utils.py) over libwseg.so (hand-written HIP for
gfx950, C-ABI in include/wseg.h).  See DESIGN.md.
"""
from .utils import RATIO_DECODING_TIME_STEP_TO_SPEC_TIME_STEP  # noqa: F401

__all__ = ["RATIO_DECODING_TIME_STEP_TO_SPEC_TIME_STEP"]

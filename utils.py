"""Drop-in import shim for `from utils import RATIO_DECODING_TIME_STEP_TO_SPEC_TIME_STEP`."""
from whisperseg_amd.utils import RATIO_DECODING_TIME_STEP_TO_SPEC_TIME_STEP  # noqa: F401

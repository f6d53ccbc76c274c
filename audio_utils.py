"""Drop-in import shim for `from audio_utils import WhisperSegFeatureExtractor, get_n_fft_given_sr`."""
from whisperseg_amd.audio_utils import WhisperSegFeatureExtractor, get_n_fft_given_sr  # noqa: F401

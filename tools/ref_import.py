"""Import the upstream reference (read-only, /root/reference) inside THIS container only.

Used exclusively by tools/make_golden.py to generate golden vectors.  Nothing under
tests/, bench.py or the package imports this module: /root/reference does not exist on
the GPU box.  Recipe follows SURVEY.md §8(c): import transformers first, then stub the
five absent third-party modules the reference imports but never uses on the hot path.
"""
import sys
import types

REFERENCE_ROOT = "/root/reference"


def import_reference():
    from transformers import (  # noqa: F401  (must precede the stubs)
        WhisperFeatureExtractor, WhisperTokenizer, WhisperForConditionalGeneration, WhisperConfig)
    from transformers.audio_utils import mel_filter_bank  # noqa: F401
    for name in ("ipywidgets", "ctranslate2", "librosa", "mutagen", "soundfile"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["mutagen"].File = None
    sys.modules["ipywidgets"].interact = None
    sys.modules["ipywidgets"].fixed = None
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import audio_utils as ref_audio_utils
    import model as ref_model
    return ref_audio_utils, ref_model

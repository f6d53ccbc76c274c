"""Drop-in import shim: `from model import WhisperSegmenter, WhisperSegmenterFast` (as reference
scripts/segment.py:9, evaluate.py and the services do) resolves to the MI355X implementation."""
from whisperseg_amd.model import (SegmenterBase, WhisperSegmenter, WhisperSegmenterFast,  # noqa: F401
                                  WhisperSegmenterForEval)

"""Drop-in import shim for `from evaluate import evaluate, evaluate_dataset` (reference evaluate.py)."""
from whisperseg_amd.evaluate import evaluate, evaluate_dataset  # noqa: F401

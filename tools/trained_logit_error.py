"""First-step logit error of engine modes against the exact-parity f32 mode on the TRAINED fixture models (tests/golden/tiny_model2,
tiny_model3) over the windows of a few sweep recordings (needs the GPU):

    python tools/trained_logit_error.py [--recordings 24] MODE [MODE ...]          e.g. f16x3 bf16x3 f16m6

On seeded random weights (tools/logit_error.py) the cross-attention is diffuse and the storage format of its K / V rows does not show;
on a trained model single encoder positions carry the probability, and it does: this is the measurement behind the x3 modes' 24-bit
block-floating-point rows (r06; knobs builds select the other formats: WSEG_X3_CKV=k24 | f32 | bfp with WSEG_LIB=.../libwseg_knobs.so).
Prints one JSON line per (model, mode): max / mean |logit diff| over all windows, the logit scale, argmax agreement."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_inputs as GI  # noqa: E402
from tools import tiny_model as TM  # noqa: E402
from whisperseg_amd.model import PROMPT_TOKENS, WhisperSegmenter  # noqa: E402


def first_logits(seg, batches):
    eng, tok = seg._first_engine()
    prompt = tok.convert_tokens_to_ids(PROMPT_TOKENS)
    out = []
    for b in batches:
        out.append(eng.generate(b.to(eng.device), prompt, tok.eos_token_id, tok.pad_token_id, max_length=8, num_beams=1,
                                suppress_tokens=seg.suppress_tokens, begin_suppress_tokens=seg.begin_suppress_tokens,
                                return_first_logits=True)[2].float().cpu())
    return torch.cat(out, 0)


def measure(variant, model, first_seed, n_rec, modes):
    mdir = os.path.join(ROOT, "tests", "golden", model)
    ref = WhisperSegmenter(mdir, device="cuda", device_ids=[0], dtype="f32")
    batches = []
    for i in range(n_rec):
        audio = GI.tiny_recording(first_seed + i, 1 + i % 6, variant=variant)
        sliced = ref.get_sliced_audios_features(audio, TM.SR, 0, TM.STS, 1)
        batches.append(torch.stack([s[2] for s in sliced]))
    want = first_logits(ref, batches)
    res = {}
    for mode in modes:
        got = first_logits(WhisperSegmenter(mdir, device="cuda", device_ids=[0], dtype=mode), batches)
        d = (got - want).abs()
        res[mode] = dict(model=model, mode=mode, windows=int(want.shape[0]), max_abs_logit_err=d.max().item(), mean_abs_logit_err=d.mean().item(),
                         logit_scale=want.abs().max().item(), argmax_equal=int((got.argmax(-1) == want.argmax(-1)).sum().item()))
        print(json.dumps(res[mode]), flush=True)
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--recordings", type=int, default=24)
    ap.add_argument("modes", nargs="+")
    a = ap.parse_args()
    measure("tiny2", "tiny_model2", 5000, a.recordings, a.modes)
    measure("tiny3", "tiny_model3", 11000, a.recordings, a.modes)

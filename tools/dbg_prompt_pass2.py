"""Debug: run 76 of the parity sweep (seed 1019) at token level: f32 vs f16x3 with / without the prompt pass."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_inputs as GI  # noqa: E402
from tools import tiny_model as TM  # noqa: E402
from tools.parity_sweep import MODEL_DIR  # noqa: E402
from whisperseg_amd.engine import Engine  # noqa: E402
from whisperseg_amd.model import WhisperSegmenter  # noqa: E402

with open(os.path.join(ROOT, "tests", "golden", "tiny_sweep.json")) as f:
    sweep = json.load(f)
run = sweep[int(os.environ.get("RUN", "76"))]
calls = []
orig = Engine.generate


def spy(self, feats, *a, **kw):
    calls.append((feats.clone(), a, dict(kw)))
    return orig(self, feats, *a, **kw)


Engine.generate = spy
segs = {dt: WhisperSegmenter(MODEL_DIR, device="cuda", device_ids=[0], dtype=dt) for dt in sys.argv[1:] or ["f32", "f16x3", "bf16x3", "f16m6"]}
first = next(iter(segs.values()))
first.segment(GI.tiny_recording(run["seed"], run["n_windows"]), TM.SR, **run["kwargs"])
Engine.generate = orig
feats, a, kw = calls[0]
print("windows", feats.shape[0], "kw", {k: v for k, v in kw.items() if k not in ("suppress_tokens", "begin_suppress_tokens")})
kw["return_first_logits"] = True
res = {}
for dt, seg in segs.items():
    eng = seg._first_engine()[0]
    for mode in ("pass", "step"):
        if mode == "step":
            os.environ["WSEG_NO_PROMPT_PASS"] = "1"
        else:
            os.environ.pop("WSEG_NO_PROMPT_PASS", None)
        t, l, fl = eng.generate(feats, *a, **kw)
        res[dt, mode] = (t.cpu(), l.cpu(), fl.float().cpu())
rt, rl, rf = res[next(iter(segs)), "step"]
for (dt, mode), (t, l, fl) in res.items():
    diff = [(i, int((t[i] != rt[i]).nonzero()[0])) for i in range(t.shape[0]) if not torch.equal(t[i], rt[i])]
    print(dt, mode, "first-logit err", float((fl - rf).abs().max()), "scale", float(rf.abs().max()), "lens", l.tolist(), "first diff (window, position)", diff)
    for i, p in diff:
        print("    window", i, "ref", rt[i, p - 2:p + 3].tolist(), "got", t[i, p - 2:p + 3].tolist())

"""Stress for rare lane-dependent results on the tiny trained model (the configuration of tests/test_scheduler_gpu.py):
python tools/lanes_stress.py [--dtype f32] [--iters 60]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from tools import tiny_model as TM
from safetensors.torch import load_file
from whisperseg_amd.engine import Engine
from whisperseg_amd.audio_utils import WhisperSegFeatureExtractor

ap = argparse.ArgumentParser(); ap.add_argument("--dtype", default="f32"); ap.add_argument("--iters", type=int, default=60)
a = ap.parse_args()
MODEL_DIR = os.path.join(ROOT, "tests", "golden", "tiny_model")
sd = {k: v.float() for k, v in load_file(os.path.join(MODEL_DIR, "model.safetensors")).items()}
cfg = json.load(open(os.path.join(MODEL_DIR, "config.json")))
eng = Engine.from_state_dict(sd, cfg, "cuda:0", a.dtype)
ext = WhisperSegFeatureExtractor(TM.SR, TM.STS, device="cuda:0")
xs = []
for s in range(23):
    clip = TM.synth_clip(np.random.default_rng(300 + s))[0]
    xs.append(ext.extract_windows(torch.from_numpy(clip).cuda(), torch.tensor([0]), len(clip))[0])
x = torch.stack(xs)


def gen(**kw):
    t, l = eng.generate(x, TM.PROMPT, TM.EOT, TM.EOT, max_length=448, num_beams=4, suppress_tokens=TM.SUPPRESS,
                        begin_suppress_tokens=TM.BEGIN_SUPPRESS, **kw)
    return t.cpu(), l.cpu()


ref = gen(n_slots=5, n_lanes=1)
bad = 0
for it in range(a.iters):
    for lanes in (1, 2, 3, 4):
        t, l = gen(n_slots=5, n_lanes=lanes)
        if not (torch.equal(t, ref[0]) and torch.equal(l, ref[1])):
            bad += 1
            rows = (t != ref[0]).any(1).nonzero().flatten().tolist()
            first = [(r, int((t[r] != ref[0][r]).nonzero()[0]), int(l[r]), int(ref[1][r])) for r in rows[:6]]
            print(f"iter {it} lanes {lanes}: windows {rows} differ; (window, first differing position, length, ref length) {first}", flush=True)
print(f"{a.dtype}: {bad} mismatching calls of {a.iters * 4}")

"""Diagnostic: where do multi-lane tokens differ from single-lane tokens at large geometry?"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whisperseg_amd.engine import Engine

ap = argparse.ArgumentParser()
ap.add_argument("--windows", type=int, default=512)
ap.add_argument("--slots", type=int, default=256)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--layers", type=int, default=32)
a = ap.parse_args()
cfg = dict(d_model=1280, encoder_attention_heads=20, decoder_attention_heads=20, encoder_layers=a.layers, decoder_layers=a.layers,
           encoder_ffn_dim=5120, decoder_ffn_dim=5120, vocab_size=51865, num_mel_bins=80, max_source_positions=500, max_target_positions=448)
eng = Engine.random(cfg, "cuda:0", a.dtype)
W = a.windows
feats = torch.randn(W, 80, 1000, device="cuda") * 0.5
enc = torch.cat([eng.encode(feats[i:i + 64]) for i in range(0, W, 64)])
prompt, eos = [50258, 50259, 50363], 50257
kw = dict(max_length=35, num_beams=4, suppress_tokens=[eos, 1, 2], begin_suppress_tokens=[220], n_slots=a.slots, encoder_output=enc)


def run(lanes):
    t, l = eng.generate(feats, prompt, eos, eos, n_lanes=lanes, **kw)
    torch.cuda.synchronize()
    return t.cpu(), l.cpu()


def diff(x, y, name):
    bad = (x[0] != y[0]).any(1)
    idx = bad.nonzero().flatten().tolist()
    first = [(i, int((x[0][i] != y[0][i]).nonzero()[0])) for i in idx[:6]]
    print(f"{name}: {len(idx)} of {W} windows differ; first windows {idx[:12]}; (window, first differing position) {first}", flush=True)


a1, a2 = run(1), run(1)
diff(a1, a2, "1 lane vs 1 lane")
b1, b2 = run(2), run(2)
diff(b1, b2, "2 lanes vs 2 lanes")
diff(a1, b1, "1 lane vs 2 lanes")
# each half on its own through one lane: a fresh call per half
h0 = eng.generate(feats[:W // 2], prompt, eos, eos, n_lanes=1, **{**kw, "encoder_output": enc[:W // 2]})
h1 = eng.generate(feats[W // 2:], prompt, eos, eos, n_lanes=1, **{**kw, "encoder_output": enc[W // 2:]})
hh = (torch.cat([h0[0], h1[0]]).cpu(), torch.cat([h0[1], h1[1]]).cpu())
diff(a1, hh, "1 lane vs two separate calls")
diff(b1, hh, "2 lanes vs two separate calls")

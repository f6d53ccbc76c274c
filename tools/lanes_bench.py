"""Throughput of wseg_generate over lanes (independent slot groups stepping side by side on their own streams).
    python tools/lanes_bench.py [--windows 1024] [--slots 256] [--lanes 1,2,3,4] [--decode-only] [--varied]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whisperseg_amd.engine import Engine

ap = argparse.ArgumentParser()
ap.add_argument("--windows", type=int, default=1024)
ap.add_argument("--slots", type=int, default=256)
ap.add_argument("--lanes", default="1,2,3,4")
ap.add_argument("--gen", type=int, default=32)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--decode-only", action="store_true", help="feed precomputed encoder states (times cross-K/V + decode)")
ap.add_argument("--refill", type=int, default=0, help="refill_min (0: slots / 8)")
ap.add_argument("--varied", action="store_true", help="per-window length caps 8..gen (exercises the refill)")
a = ap.parse_args()
cfg = dict(d_model=1280, encoder_attention_heads=20, decoder_attention_heads=20, encoder_layers=32, decoder_layers=32,
           encoder_ffn_dim=5120, decoder_ffn_dim=5120, vocab_size=51865, num_mel_bins=80, max_source_positions=500, max_target_positions=448)
eng = Engine.random(cfg, "cuda:0", a.dtype)
W = a.windows
feats = torch.randn(W, 80, 1000, device="cuda") * 0.5
enc = torch.cat([eng.encode(feats[i:i + 64]) for i in range(0, W, 64)]) if a.decode_only else None
prompt, eos = [50258, 50259, 50363], 50257
kw = dict(max_length=3 + a.gen, num_beams=4, suppress_tokens=[eos, 1, 2], begin_suppress_tokens=[220], n_slots=a.slots, refill_min=a.refill)
if a.varied:
    g = torch.Generator().manual_seed(0)
    kw["window_max_length"] = torch.randint(3 + 8, 3 + a.gen + 1, (W,), generator=g, dtype=torch.int32)
ref = None
for lanes in [int(v) for v in a.lanes.split(",")]:
    best = None
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        toks, lens = eng.generate(feats, prompt, eos, eos, encoder_output=enc, n_lanes=lanes, **kw)
        torch.cuda.synchronize(); dt = (time.time() - t0) * 1e3
        best = dt if best is None else min(best, dt)
    st = eng.last_stats()
    tm = eng.last_timing()
    same = ""
    if ref is None:
        ref = (toks.clone(), lens.clone())
    else:
        same = f" tokens equal to 1 lane: {torch.equal(ref[0], toks) and torch.equal(ref[1], lens)}"
    print(f"lanes {lanes} x {st['n_slots'] // st['n_lanes']} slots, {W} windows: {best:8.1f} ms  {W * 30 / best * 1e3:8.0f} audio-s/s | "
          f"steps {st['n_steps']} admissions {st['n_admissions']} occupancy {st['occupancy']:.2f} | per-lane enc {tm[0]:.0f} ckv {tm[1]:.0f} ms{same}", flush=True)

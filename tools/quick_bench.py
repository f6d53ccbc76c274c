"""Ad-hoc timing probe (not the contract bench): python tools/quick_bench.py --model large --windows 8"""
import argparse, time, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whisperseg_amd.engine import Engine

GEO = {"large": dict(d=1280, h=20, L=32, f=5120), "base": dict(d=512, h=8, L=6, f=2048), "tiny": dict(d=128, h=2, L=2, f=512)}
ap = argparse.ArgumentParser()
ap.add_argument("--model", default="large"); ap.add_argument("--windows", type=int, default=8)
ap.add_argument("--gen", type=int, default=32); ap.add_argument("--beams", type=int, default=4)
ap.add_argument("--iters", type=int, default=3); ap.add_argument("--dtype", default="bf16")
ap.add_argument("--slots", type=int, default=0, help="window slots (default: one per window)")
ap.add_argument("--decode-only", action="store_true", help="encoder states precomputed (64 windows at a time): the call is cross-K/V + decode")
a = ap.parse_args()
g = GEO[a.model]
cfg = dict(d_model=g["d"], encoder_attention_heads=g["h"], decoder_attention_heads=g["h"], encoder_layers=g["L"], decoder_layers=g["L"],
           encoder_ffn_dim=g["f"], decoder_ffn_dim=g["f"], vocab_size=51865, num_mel_bins=80, max_source_positions=500, max_target_positions=448)
eng = Engine.random(cfg, "cuda:0", a.dtype)
feats = torch.randn(a.windows, 80, 1000, device="cuda") * 0.5
prompt, eos = [50258, 50259, 50363], 50257
extra = {}
if a.decode_only:
    extra["encoder_output"] = torch.cat([eng.encode(feats[lo:lo + 64]) for lo in range(0, a.windows, 64)])
for it in range(a.iters):
    torch.cuda.synchronize(); t0 = time.time()
    toks, lens = eng.generate(feats, prompt, eos, eos, max_length=3 + a.gen, num_beams=a.beams, suppress_tokens=[eos, 1, 2], begin_suppress_tokens=[220],
                                n_slots=a.slots or a.windows, **extra)
    torch.cuda.synchronize(); dt = time.time() - t0
    enc, ckv, dec, steps = eng.last_timing()
    print(f"iter {it}: total {dt*1e3:.1f} ms | enc {enc:.1f} ckv {ckv:.1f} dec {dec:.1f} ms over {int(steps)} steps ({dec/max(steps,1):.3f} ms/step) | {a.windows/dt:.1f} windows/s", flush=True)
print("lens", lens.tolist()[:8])

"""Encoder-only timing (large geometry): python tools/enc_attn_bench.py [--windows 256]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whisperseg_amd.engine import Engine
ap = argparse.ArgumentParser(); ap.add_argument("--windows", type=int, default=256); ap.add_argument("--dtype", default="bf16"); ap.add_argument("--layers", type=int, default=32); a = ap.parse_args()
cfg = dict(d_model=1280, encoder_attention_heads=20, decoder_attention_heads=20, encoder_layers=a.layers, decoder_layers=1,
           encoder_ffn_dim=5120, decoder_ffn_dim=5120, vocab_size=51865, num_mel_bins=80, max_source_positions=500,
           max_target_positions=448)
eng = Engine.random(cfg, "cuda:0", a.dtype, seed=0)
x = torch.randn(a.windows, 80, 1000, device="cuda") * 0.5
for _ in range(2): eng.encode(x)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(3): eng.encode(x)
e1.record(); torch.cuda.synchronize()
print("encoder %.2f ms per %d windows" % (e0.elapsed_time(e1) / 3, a.windows))

"""Probe: does the decode phase gain from two slot groups stepping concurrently on two streams?
A decode step alternates latency-bound GEMM launches (320 workgroups) with HBM-bound attention launches; two
independent groups on two streams could fill each other's gaps.  Two engines (same weights), two host threads,
precomputed encoder states (wseg_generate's encoder_output hook), wall time against one engine doing the same windows.
    python tools/dual_decode_probe.py [--windows 256]"""
import argparse, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whisperseg_amd.engine import Engine

ap = argparse.ArgumentParser()
ap.add_argument("--windows", type=int, default=256)
ap.add_argument("--gen", type=int, default=32)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--groups", type=int, default=2)
a = ap.parse_args()
cfg = dict(d_model=1280, encoder_attention_heads=20, decoder_attention_heads=20, encoder_layers=32, decoder_layers=32,
           encoder_ffn_dim=5120, decoder_ffn_dim=5120, vocab_size=51865, num_mel_bins=80, max_source_positions=500, max_target_positions=448)
G = a.groups
engs = [Engine.random(cfg, "cuda:0", a.dtype) for _ in range(G)]
W = a.windows
feats = torch.randn(W, 80, 1000, device="cuda") * 0.5
enc = torch.cat([engs[0].encode(feats[i:i + 64]) for i in range(0, W, 64)])
prompt, eos = [50258, 50259, 50363], 50257
kw = dict(max_length=3 + a.gen, num_beams=4, suppress_tokens=[eos, 1, 2], begin_suppress_tokens=[220])


def run(e, lo, hi, out, k, stream):
    with torch.cuda.stream(stream):
        out[k] = e.generate(feats[lo:hi], prompt, eos, eos, encoder_output=enc[lo:hi], **kw)
        stream.synchronize()


def serial(n):
    out = {}
    torch.cuda.synchronize(); t0 = time.time()
    run(engs[0], 0, n, out, 0, torch.cuda.current_stream())
    torch.cuda.synchronize()
    return (time.time() - t0) * 1e3, out


def dual(per):
    out = {}
    streams = [torch.cuda.Stream() for _ in range(G)]
    th = [threading.Thread(target=run, args=(engs[g], (g * per) % W, (g * per) % W + per, out, g, streams[g])) for g in range(G)]
    torch.cuda.synchronize(); t0 = time.time()
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    return (time.time() - t0) * 1e3, out


for it in range(3):
    ts, o1 = serial(W)
    print(f"one group of {W}: {ts:.1f} ms (decode {engs[0].last_timing()[2]:.1f} ms)", flush=True)
for it in range(3):
    td, o2 = dual(W // G)
    print(f"{G} concurrent groups of {W // G}: {td:.1f} ms", flush=True)
same = all(torch.equal(o1[0][0][g * (W // G):(g + 1) * (W // G)], o2[g][0]) for g in range(G))
print("tokens equal to the single group:", same)
if W * G <= 1024:
    for it in range(2):
        td, _ = dual(W)
        print(f"{G} concurrent groups of {W}: {td:.1f} ms  (serial would be {G * ts:.1f})", flush=True)

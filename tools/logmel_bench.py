"""Front-end timing over the reference's window configurations: python tools/logmel_bench.py [--windows 256]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whisperseg_amd.audio_utils import get_feature_extractor
ap = argparse.ArgumentParser(); ap.add_argument("--windows", type=int, default=256); a = ap.parse_args()
for (sr, sts) in ((16000, 0.03), (16000, 0.01), (32000, 0.0025), (48000, 0.0025)):
    W = a.windows
    wl = int(1000 * sts * sr)
    ext = get_feature_extractor(sr, sts, 0, 30, 1000, "cuda:0")
    audio = torch.randn(W * wl, device="cuda") * 0.1
    st = (torch.arange(W, dtype=torch.int64) * wl).cuda()
    ext.extract_windows(audio, st, wl)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ext.extract_windows(audio, st, wl)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    b = W * (4 * wl + 320000)
    print(f"sr {sr} sts {sts} n_fft {ext.n_fft} hop {ext.hop_length}: {ms:.3f} ms  {b / ms / 1e6:.0f} GB/s algorithmic ({100 * b / ms / 1e6 / 8000:.1f} % of 8 TB/s)", flush=True)

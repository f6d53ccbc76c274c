"""Calibration only (never a product path): torch's scaled_dot_product_attention backends on the encoder attention shape
[windows, 20 heads, 500 positions, 64] bf16, non-causal, against which enc_attention_h16_kernel's 767 us per layer
(427 TFLOP/s at 256 windows) can be read.   python tools/sdpa_calib.py [--windows 256]"""
import argparse
import torch
import torch.nn.functional as F
from torch.nn.attention import SDPBackend, sdpa_kernel

ap = argparse.ArgumentParser(); ap.add_argument("--windows", type=int, default=256); a = ap.parse_args()
W, H, T, D = a.windows, 20, 500, 64
q, k, v = (torch.randn(W, H, T, D, device="cuda", dtype=torch.bfloat16) for _ in range(3))
flops = 4.0 * T * T * D * H * W
for name, be in (("flash", SDPBackend.FLASH_ATTENTION), ("efficient", SDPBackend.EFFICIENT_ATTENTION), ("math", SDPBackend.MATH)):
    try:
        with sdpa_kernel(be):
            for _ in range(3):
                F.scaled_dot_product_attention(q, k, v)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(10):
                F.scaled_dot_product_attention(q, k, v)
            e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        print(f"sdpa {name:10s}: {us:8.1f} us  {flops / us / 1e6:7.1f} TFLOP/s", flush=True)
    except Exception as exc:
        print(f"sdpa {name:10s}: unavailable ({type(exc).__name__}: {str(exc)[:100]})", flush=True)

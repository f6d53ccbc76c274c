"""Shader-cycle stamps of one wave of logmel_fft_kernel over one frame (measurement build: python -m whisperseg_amd.build --stamps 5;
WSEG_LIB=whisperseg_amd/lib/libwseg_stamps5.so python tools/logmel_stamps.py): load + window | FFT | unpack + power | mel items |
filter sums + log.  With three workgroups per CU a wave shares its SIMD with two others: the spans are ELAPSED cycles and include
their turns (a stamp itself costs ~100)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whisperseg_amd import _lib
from whisperseg_amd.audio_utils import get_feature_extractor

W, sr, sts = 256, 16000, 0.01
wl = int(1000 * sts * sr)
ext = get_feature_extractor(sr, sts, 0, 30, 1000, "cuda:0")
audio = torch.randn(W * wl, device="cuda") * 0.1
st = (torch.arange(W, dtype=torch.int64) * wl).cuda()
raw = ctypes.CDLL(_lib.LIB_PATH)
names = ["load + window", "FFT", "unpack + power", "mel items", "filter sums + log"]
for it in range(3):
    for _ in range(5):
        ext.extract_windows(audio, st, wl)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 16)()
    raw.wseg_debug_logmel_stamps(buf)
    t = [buf[i] for i in range(6)]
    print("  ".join(f"{n} {t[i + 1] - t[i]}" for i, n in enumerate(names)) + f"  | frame {t[5] - t[0]} cycles", flush=True)

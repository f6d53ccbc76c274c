"""Cycle stamps of the 256x256 ping-pong GEMM kernel (measurement build: python -m whisperseg_amd.build --stamps 4, then on the GPU box
WSEG_LIB=whisperseg_amd/lib/libwseg_stamps4.so python tools/pp_stamps.py [--shape M,N,K,epi]).  Workgroup 0, wave 0 (row group 0)
and wave 4 (row group 1), K tiles 8..11: per phase (A, B) the L part (fragment ds_reads + LDS-DMA issue), the wait (lgkmcnt(0) +
barrier), the M part (32 MFMAs; 48 in the split-precision modes), in shader cycles (s_memtime).  A stamp itself costs ~100 cycles
(s_memtime returns through the scalar data path): subtract it from every span before reading absolute numbers."""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whisperseg_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="128000,1280,1280,2")
ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16x3", "bf16x3", "f16m6"], help="f16m6: K tiles alternate hi (even) / MX (odd)")
ap.add_argument("--iters", type=int, default=40, help="launches before the stamps are read (the clock settles within a few ms of load)")
a = ap.parse_args()
m, n, k, epi = (int(v) for v in a.shape.split(","))
lib = _lib.load(require_device=True)
raw = ctypes.CDLL(_lib.LIB_PATH)
from whisperseg_amd.engine import DTYPES, SPLIT_BASE, split_operand
DT = DTYPES[a.dtype][0]
if a.dtype == "bf16":
    A = (torch.rand(m, k, device="cuda") * 2 - 1).to(torch.bfloat16)
    W = (torch.rand(n, k, device="cuda") * 2 - 1).to(torch.bfloat16)
    pd = torch.bfloat16
else:
    A = split_operand(torch.rand(m, k, device="cuda") * 2 - 1, SPLIT_BASE[a.dtype])
    W = split_operand((torch.rand(n, k, device="cuda") * 2 - 1) * k ** -0.5, SPLIT_BASE[a.dtype])
    if a.dtype == "f16m6":
        Am, Wm = torch.empty_like(A), torch.empty_like(W)
        _lib.check(lib.wseg_convert_operand(A.data_ptr(), Am.data_ptr(), m, k, 0, _lib.stream_ptr()))
        _lib.check(lib.wseg_convert_operand(W.data_ptr(), Wm.data_ptr(), n, k, 1, _lib.stream_ptr()))
        A, W = Am, Wm
    pd = torch.float32
bias = torch.rand(n, device="cuda").to(pd)
od = torch.float32 if epi == 2 else pd
res = torch.rand(m, n, device="cuda").to(od)
out = torch.empty(m, n, device="cuda", dtype=od)
ws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
for it in range(a.iters):
    _lib.check(lib.wseg_debug_gemm(DT, epi, m, n, k, A.data_ptr(), W.data_ptr(), bias.data_ptr(), res.data_ptr(), out.data_ptr(), ws.data_ptr(),
                                   ws.numel(), _lib.stream_ptr()))
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 132)()
raw.wseg_debug_pp_stamps(buf)
S = [[[[buf[((g * 4 + t) * 4 + p) * 4 + i] for i in range(4)] for p in range(4)] for t in range(4)] for g in range(2)]
print(f"shape M={m} N={n} K={k} epi={epi} {a.dtype}")
for g in range(2):
    for t in range(1, 3):
        row = []
        for p in range(2):
            s = S[g][t][p]
            nxt = S[g][t][p + 1][0] if p < 1 else S[g][t + 1][0][0]
            row.append(f"phase {'AB'[p]}: L {s[1] - s[0]:4d} wait {s[2] - s[1]:4d} M {s[3] - s[2]:4d} tail {nxt - s[3]:4d}")
        print(f"group {g} K tile {8 + t}: " + " | ".join(row) + f" | K tile {S[g][t + 1][0][0] - S[g][t][0][0]} cycles")
t0 = S[0][1][0][0]
print("group 1 lags group 0 by", S[1][1][0][0] - t0, "cycles at the top of K tile 9")
dc, dt = buf[130] - buf[128], (buf[131] - buf[129]) / 100.0
print(f"K tiles 2..18 of workgroup 0: {dc} shader cycles in {dt:.2f} us = {dc / dt / 1e3:.3f} GHz effective clock, {dt / 16:.3f} us per K tile (stamped workgroup)")

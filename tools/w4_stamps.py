"""Cycle stamps of the one-wave-per-SIMD GEMM kernel (measurement build: python -m whisperseg_amd.build --stamps 6 -DWSEG_KNOBS=1, then on the
GPU box  WSEG_LIB=whisperseg_amd/lib/libwseg_stamps6.so WSEG_GEMM_W4=1 python tools/w4_stamps.py [--shape M,N,K,epi] [--dtype bf16|f16m6]).
Workgroup 0, all four waves: per K tile the first half (MFMAs + fragment reads, issue time), the wait (vmcnt(0) lgkmcnt(0) + barrier), the second
half (MFMAs + reads + LDS-DMA issue); and the epilogue.  A stamp costs ~50-100 cycles (s_memtime + a store)."""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whisperseg_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="128000,1280,1280,2")
ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16m6"])
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--waves", default="0,3")
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
    Am, Wm = torch.empty_like(A), torch.empty_like(W)
    _lib.check(lib.wseg_convert_operand(A.data_ptr(), Am.data_ptr(), m, k, 0, _lib.stream_ptr()))
    _lib.check(lib.wseg_convert_operand(W.data_ptr(), Wm.data_ptr(), n, k, 1, _lib.stream_ptr()))
    A, W = Am, Wm
    pd = torch.float32
bias = torch.rand(n, device="cuda").to(pd)
od = torch.float32 if epi == 2 else pd
res = torch.rand(m, n, device="cuda").to(od)
out = torch.empty(m, n if (epi == 2 or a.dtype == "bf16") else 2 * n, device="cuda", dtype=od if (epi == 2 or a.dtype == "bf16") else torch.int16)
ws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
for it in range(a.iters):
    _lib.check(lib.wseg_debug_gemm(DT, epi, m, n, k, A.data_ptr(), W.data_ptr(), bias.data_ptr(), res.data_ptr(), out.data_ptr(), ws.data_ptr(),
                                   ws.numel(), _lib.stream_ptr()))
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (4 * 512 + 32))()
raw.wseg_debug_w4_stamps(buf)
NAMES = {1: "hi/plain tile: top", 2: "first half issued", 3: "past mid barrier", 4: "second half issued", 5: "MX tile: top", 6: "first half issued",
         7: "past mid barrier", 8: "second half issued", 9: "epilogue begins", 10: "epilogue ends"}
print(f"shape M={m} N={n} K={k} epi={epi} {a.dtype}")
for w in (int(x) for x in a.waves.split(",")):
    ev = [(buf[w * 512 + i] >> 4, buf[w * 512 + i] & 15) for i in range(512) if buf[w * 512 + i]]
    if not ev:
        continue
    t0 = ev[0][0]
    print(f"wave {w}: {len(ev)} stamps")
    line, last = [], t0
    tiles = 0
    for t, tag in ev:
        if tag in (1, 5, 9):
            if line:
                print("   " + "  ".join(line))
            line = [f"@{t - t0:7d}"]
            tiles += 1
        line.append(f"{ {1: 'T', 2: 'h1', 3: 'bar', 4: 'h2', 5: 'X', 6: 'x1', 7: 'bar', 8: 'x2', 9: 'EPI', 10: 'end', 11: 'lgkm', 12: 'vm'}[tag]}+{t - last}")
        last = t
        if tiles > 50:
            break
    if line:
        print("   " + "  ".join(line))

pairs = [(buf[4 * 512 + 2 * e], buf[4 * 512 + 2 * e + 1]) for e in range(16) if buf[4 * 512 + 2 * e]]
if len(pairs) >= 2:
    dc, dt = pairs[-1][0] - pairs[0][0], (pairs[-1][1] - pairs[0][1]) / 100.0
    print(f"wave 0, epilogue {0} .. {len(pairs) - 1}: {dc} shader cycles in {dt:.2f} us = {dc / dt / 1e3:.3f} GHz effective clock; {dt / (len(pairs) - 1):.2f} us per output tile")

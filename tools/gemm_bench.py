"""GEMM tuning probe: TFLOP/s of libwseg's bf16 GEMM on the encoder / decoder shapes (random [-1,1) data).
    python tools/gemm_bench.py [--windows 32]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whisperseg_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--windows", type=int, default=32)
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--encoder-only", action="store_true")
ap.add_argument("--decoder-only", action="store_true")
ap.add_argument("--rotate", type=int, default=1, help="cycle through this many weight copies (cold weights, as in a decode step)")
ap.add_argument("--vs-torch", action="store_true", help="also time torch's F.linear (hipBLASLt) on the same operands: calibration of what the box can do, never a product path")
ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16x3", "bf16x3", "f16m6"], help="engine mode of the GEMM (split modes: operands from fp32 values; f16m6: M6 rows via wseg_convert_operand)")
ap.add_argument("--shapes", default="", help="custom list 'M,N,K,epi;M,N,K,epi;...' (epi 0 bias, 1 bias+gelu, 2 bias+residual, 3 the decoder's fused bias+residual+LayerNorm step)")
a = ap.parse_args()
lib = _lib.load(require_device=True)
d, f = 1280, 5120
M = a.windows * 500
shapes = [("qkv", M, 3 * d, d, 0), ("o-proj", M, d, d, 2), ("fc1", M, f, d, 1), ("fc2", M, d, f, 2), ("conv2", M, d, 3 * d, 1),
          ("4096^3", 4096, 4096, 4096, 0), ("dec fc1 R=128", 128, f, d, 1), ("dec fc2 R=128", 128, d, f, 2), ("dec o R=128", 128, d, d, 2),
          ("dec fc1 R=480", 480, f, d, 1), ("dec fc2 R=480", 480, d, f, 2), ("dec o R=480", 480, d, d, 2), ("dec qkv R=480", 480, 3 * d, d, 0),
          ("dec fc1 R=32", 32, f, d, 1), ("dec fc2 R=32", 32, d, f, 2), ("dec qkv R=32", 32, 3 * d, d, 0), ("lm head R=128", 128, 51968, d, 0)]
if a.encoder_only:
    shapes = shapes[:5]
if a.decoder_only:
    shapes = [sh for sh in shapes if sh[0].startswith('dec') or sh[0].startswith('lm')]
if a.shapes:
    shapes = [("custom",) + tuple(int(v) for v in item.split(",")) for item in a.shapes.split(";") if item]
ws = torch.empty(512 << 20, dtype=torch.uint8, device="cuda")
from whisperseg_amd.engine import DTYPES, SPLIT_BASE, split_operand, unsplit_m6, unsplit_operand
DT = DTYPES[a.dtype][0]
split = a.dtype != "bf16"
for name, m, n, k, epi in shapes:
    mp = (m + 255) // 256 * 256
    if not split:
        A = (torch.rand(mp, k, device="cuda") * 2 - 1).to(torch.bfloat16)
        W = (torch.rand(n, k, device="cuda") * 2 - 1).to(torch.bfloat16)
        A_ref, W_ref = A, W
        pd = torch.bfloat16
    else:      # split-precision modes: fp32 values as hi | lo operand rows (f16m6: converted to M6 rows on the device)
        A_ref = torch.rand(mp, k, device="cuda") * 2 - 1
        W_ref = (torch.rand(n, k, device="cuda") * 2 - 1) * k ** -0.5
        A, W = split_operand(A_ref, SPLIT_BASE[a.dtype]), split_operand(W_ref, SPLIT_BASE[a.dtype])
        if a.dtype == "f16m6":
            Am, Wm = torch.empty_like(A), torch.empty_like(W)
            _lib.check(lib.wseg_convert_operand(A.data_ptr(), Am.data_ptr(), mp, k, 0, _lib.stream_ptr()))
            _lib.check(lib.wseg_convert_operand(W.data_ptr(), Wm.data_ptr(), n, k, 1, _lib.stream_ptr()))
            A, W = Am, Wm
        pd = torch.float32
    Ws = [W] + [W.clone() for _ in range(a.rotate - 1)]
    call = [0]
    bias = torch.rand(n, device="cuda").to(pd)
    od = torch.float32 if epi >= 2 else pd       # the residual-stream epilogue reads / writes fp32
    gam, bet = torch.rand(n, device="cuda").to(pd), torch.rand(n, device="cuda").to(pd)
    y = torch.empty(mp, n, device="cuda", dtype=pd)
    res = torch.rand(mp, n, device="cuda").to(od)
    out = torch.empty(mp, n, device="cuda", dtype=od)
    st = _lib.stream_ptr()

    def run():
        call[0] += 1
        if epi == 3:
            _lib.check(lib.wseg_debug_gemm_resid_ln(DT, m, n, k, A.data_ptr(), Ws[call[0] % len(Ws)].data_ptr(), bias.data_ptr(), out.data_ptr(),
                                                    gam.data_ptr(), bet.data_ptr(), y.data_ptr(), ws.data_ptr(), ws.numel(), st))
            return
        _lib.check(lib.wseg_debug_gemm(DT, epi, m, n, k, A.data_ptr(), Ws[call[0] % len(Ws)].data_ptr(), bias.data_ptr(), res.data_ptr(),
                                       out.data_ptr(), ws.data_ptr(), ws.numel(), st))
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(a.iters):
        run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / a.iters * 1e3
    if epi == 3:
        out.copy_(res); run()
    ref = (A_ref[:m].float() @ W_ref.float().T + bias.float())
    if epi == 1: ref = torch.nn.functional.gelu(ref)
    if epi >= 2: ref = ref + res[:m].float()
    got = out[:m]
    if split and epi < 2:      # operand rows: hi | lo pairs, or M6 rows from the large-tile kernels of f16m6
        rows = out[:m].view(torch.int16).view(m, 2 * n)
        got = unsplit_m6(rows) if lib.wseg_debug_gemm_out_is_mx(DT, m, n, k) else unsplit_operand(rows, SPLIT_BASE[a.dtype])
    err = (got.float() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-9)
    es = 4 if split else 2
    gb = (m * k + n * k + m * n * (2 if epi == 2 else 1)) * es / 1e9
    lt = ""
    if a.vs_torch:
        Am = A_ref[:m].to(torch.bfloat16)
        Ws = [w.to(torch.bfloat16) if not split else W_ref.to(torch.bfloat16) for w in Ws]
        bias = bias.to(torch.bfloat16)
        for _ in range(3):
            torch.nn.functional.linear(Am, Ws[0], bias)
        torch.cuda.synchronize(); e0.record()
        for i in range(a.iters):
            torch.nn.functional.linear(Am, Ws[i % len(Ws)], bias)
        e1.record(); torch.cuda.synchronize()
        ut = e0.elapsed_time(e1) / a.iters * 1e3
        lt = f"  | torch F.linear (bias only) {ut:8.1f} us {2*m*n*k/ut/1e6:7.1f} TFLOP/s"
    print(f"{name:16s} M={m:6d} N={n:6d} K={k:5d}: {us:8.1f} us  {2*m*n*k/us/1e6:7.1f} TFLOP/s  {gb/us*1e6/1e3:6.2f} TB/s  relerr {err:.1e}{lt}", flush=True)

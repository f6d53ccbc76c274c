"""Stress: is the library GEMM bit-stable while another stream hammers HBM / the CUs?
A latent missing wait (LDS-DMA landing after the fragment reads, say) can hide behind regular timing on an otherwise idle
chip and show up once a second stream changes the latencies.  Runs each shape solo (reference), then repeatedly while a hog
stream runs, and counts outputs that differ from the solo result.
    python tools/concurrency_stress.py [--hog copy|gemm|both] [--iters 200]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whisperseg_amd import _lib

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--hog", default="both")
ap.add_argument("--dtype", type=int, default=1, help="1 bf16, 2 f16, 5 f16m6 (M6 operand rows from fp32 values)")
ap.add_argument("--only", default="", help="substring filter on the shape names")
a = ap.parse_args()
lib = _lib.load(require_device=True)
d, f = 1280, 5120
shapes = [("p128 store", 8192, d, d, 0), ("p128 gelu", 8192, d, d, 1), ("p128 resid", 8192, d, d, 2), ("p128 resid K5120", 8192, d, f, 2),
          ("pp resid", 65536, d, d, 2), ("pp store", 65536, d, d, 0),
          ("dec o/cq/co", 1024, d, d, 2), ("dec qkv(part)", 1024, 3 * d, d, 0), ("dec fc1", 1024, f, d, 1), ("dec fc2", 1024, d, f, 2),
          ("lm head", 1024, 51968, d, 0), ("enc o", 8192, d, d, 2), ("enc fc1", 8192, f, d, 1)]
ws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
side = torch.cuda.Stream()
hog_a = torch.empty(1 << 28, dtype=torch.float32, device="cuda")      # 1 GiB
hog_b = torch.empty_like(hog_a)
hm = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)


def hog(n):
    with torch.cuda.stream(side):
        for i in range(n):
            if a.hog in ("copy", "both"):
                hog_b.copy_(hog_a)
            if a.hog in ("gemm", "both"):
                torch.mm(hm, hm)


for name, m, n, k, epi in shapes:
    if a.only and a.only not in name:
        continue
    mp = (m + 255) // 256 * 256
    g = torch.Generator(device="cuda").manual_seed(1)
    td = torch.bfloat16 if a.dtype == 1 else torch.float16
    if a.dtype == 5:
        from whisperseg_amd.engine import split_operand
        As = split_operand(torch.rand(mp, k, device="cuda", generator=g) * 2 - 1, torch.float16)
        Ws = split_operand((torch.rand(n, k, device="cuda", generator=g) * 2 - 1) * k ** -0.5, torch.float16)
        A, W = torch.empty_like(As), torch.empty_like(Ws)
        _lib.check(lib.wseg_convert_operand(As.data_ptr(), A.data_ptr(), mp, k, 0, _lib.stream_ptr()))
        _lib.check(lib.wseg_convert_operand(Ws.data_ptr(), W.data_ptr(), n, k, 1, _lib.stream_ptr()))
        bias = torch.rand(n, device="cuda", generator=g)
        res = torch.rand(mp, n, device="cuda", generator=g)
        out = torch.empty(mp, n, device="cuda") if epi == 2 else torch.empty(mp, 2 * n, device="cuda", dtype=torch.int16)
    else:
        A = (torch.rand(mp, k, device="cuda", generator=g) * 2 - 1).to(td)
        W = (torch.rand(n, k, device="cuda", generator=g) * 2 - 1).to(td)
        bias = torch.rand(n, device="cuda", generator=g).to(td)
        od = torch.float32 if epi == 2 else td
        res = torch.rand(mp, n, device="cuda", generator=g).to(od)
        out = torch.empty(mp, n, device="cuda", dtype=od)
    st = _lib.stream_ptr()

    def run():
        _lib.check(lib.wseg_debug_gemm(a.dtype, epi, m, n, k, A.data_ptr(), W.data_ptr(), bias.data_ptr(), res.data_ptr(),
                                       out.data_ptr(), ws.data_ptr(), ws.numel(), st))
    run(); torch.cuda.synchronize()
    ref = out[:m].clone()
    solo_bad = 0
    for _ in range(20):
        out.zero_(); run(); torch.cuda.synchronize()
        solo_bad += int(not torch.equal(out[:m], ref))
    bad = 0
    worst = 0.0
    for it in range(0, a.iters, 20):
        hog(40)
        for _ in range(20):
            out.zero_()
            run()
            cur = out[:m].clone()
            if not torch.equal(cur, ref):
                bad += 1
                worst = max(worst, (cur.float() - ref.float()).abs().max().item())
                if bad <= 6:
                    dm = (cur != ref)
                    rows = dm.any(1).nonzero().flatten()
                    cols = dm.any(0).nonzero().flatten()
                    zero = (cur[dm] == 0).float().mean().item()
                    if a.dtype != 5 and k // 64 <= 80 and len(cols) % 8 == 0 and len(cols) <= 32:
                        # which K tile's weight piece was stale, and what did the LDS hold instead?  (per 8-row weight piece)
                        r0 = int(rows[0])
                        Af = A[r0:r0 + len(rows)].float()
                        for c0 in sorted(set(int(c) // 8 * 8 for c in cols)):
                            diff = (cur[r0:r0 + len(rows), c0:c0 + 8].float() - ref[r0:r0 + len(rows), c0:c0 + 8].float())
                            best = None
                            for kt in range(k // 64):
                                At = Af[:, kt * 64:(kt + 1) * 64]
                                Wt = W[c0:c0 + 8, kt * 64:(kt + 1) * 64].float()
                                rhs = diff + At @ Wt.T
                                sol = torch.linalg.lstsq(At, rhs).solution            # [64, 8] = what the LDS held, transposed
                                resid = (At @ sol - rhs).abs().max().item()
                                if best is None or resid < best[0]:
                                    best = (resid, kt, sol.T.contiguous())
                            resid, kt, stale = best
                            kinds = []
                            for j in range(8):          # per 16-byte chunk: fresh (this K tile), old (K tile - 2, same LDS slot) or other
                                row = ""
                                for c in range(8):
                                    got = stale[j, c * 8:c * 8 + 8]
                                    fresh = W[c0 + j, kt * 64 + c * 8:kt * 64 + c * 8 + 8].float()
                                    oldv = W[c0 + j, (kt - 2) * 64 + c * 8:(kt - 2) * 64 + c * 8 + 8].float() if kt >= 2 else None
                                    if (got - fresh).abs().max().item() < 1e-2: row += "f"
                                    elif oldv is not None and (got - oldv).abs().max().item() < 1e-2: row += "o"
                                    else: row += "?"
                                kinds.append(row)
                            print(f"   piece W rows {c0}..{c0 + 7}: K tile {kt} of {k // 64} (lstsq residual {resid:.1e}), tile rows {r0}..; per row: {kinds}; "
                                  f"stale row sample {stale[0][:4].tolist()}", flush=True)
                    print(f"   mismatch: {int(dm.sum())} elements, rows {int(rows[0])}..{int(rows[-1])} ({len(rows)}), cols {int(cols[0])}..{int(cols[-1])} ({len(cols)}), "
                          f"fraction still zero {zero:.2f}; sample cur {cur[dm][:4].tolist()} ref {ref[dm][:4].tolist()}", flush=True)
        torch.cuda.synchronize()
    print(f"{name:14s} M={m} N={n} K={k} epi={epi}: solo mismatches {solo_bad}/20, under load {bad}/{a.iters} (max abs diff {worst:.3g})", flush=True)

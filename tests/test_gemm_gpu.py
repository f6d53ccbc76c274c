"""libwseg GEMM (bf16 / f16 MFMA tiles incl. the split-K skinny family, and the f32 exact kernel) vs torch fp32."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [  # M, N, K
    (500, 128, 128), (1500, 384, 128), (4000, 1280, 1280), (1000, 512, 256), (3, 384, 128), (32, 3840, 1280),
    (48, 1280, 5120), (128, 1280, 1280), (200, 5120, 1280), (512, 1280, 5120), (20, 51968, 128), (257, 256, 64),
]


@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("epi", [0, 1, 2])
@pytest.mark.parametrize("dtype", ["bf16", "f16", "f32"])
def test_gemm_matches_torch(gpu_lib, M, N, K, epi, dtype):
    from whisperseg_amd import _lib
    if dtype == "f32" and M * N * K > 3e9:
        pytest.skip("f32 exact kernel is for small problems")
    td = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[dtype]
    g = torch.Generator(device="cuda").manual_seed(M * 31 + N * 7 + K)
    Mp = (M + 255) // 256 * 256
    A = (torch.rand(Mp, K, device="cuda", generator=g) * 2 - 1).to(td)
    W = ((torch.rand(N, K, device="cuda", generator=g) * 2 - 1) * K ** -0.5).to(td)      # asymmetric operands
    bias = (torch.rand(N, device="cuda", generator=g) - 0.5).to(td)
    # epi 2 is the residual-stream epilogue: resid and out are fp32 in every mode
    od = torch.float32 if epi == 2 else td
    res = (torch.rand(Mp, N, device="cuda", generator=g) - 0.5).to(od)
    out = torch.full((Mp, N), float("nan"), device="cuda", dtype=od)
    ws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    _lib.check(gpu_lib.wseg_debug_gemm({"f32": 0, "bf16": 1, "f16": 2}[dtype], epi, M, N, K, A.data_ptr(), W.data_ptr(), bias.data_ptr(),
                                       res.data_ptr(), out.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr()))
    ref = A[:M].float() @ W.float().T + bias.float()
    if epi == 1:
        ref = torch.nn.functional.gelu(ref)
    if epi == 2:
        ref = ref + res[:M].float()
    got = out[:M].float()
    assert torch.isfinite(got).all()
    tol = {"bf16": 2e-2, "f16": 3e-3, "f32": 2e-5}[dtype]     # output rounding (2^-8 / 2^-11) dominates; f32 is accumulation order only
    assert (got - ref).abs().max().item() <= tol * max(1.0, ref.abs().max().item())
    assert torch.isnan(out[M:]).all() or M == Mp      # rows beyond M are never written


X3_SHAPES = SHAPES + [(12800, 1280, 1280), (10000, 3840, 192), (30000, 5120, 1280), (4100, 1280, 5120), (2100, 3840, 1280), (700, 256, 96)]


@pytest.mark.parametrize("M,N,K", X3_SHAPES)
@pytest.mark.parametrize("epi", [0, 1, 2])
@pytest.mark.parametrize("dtype", ["bf16x3", "f16x3"])
def test_split_precision_gemm_matches_fp64(gpu_lib, M, N, K, epi, dtype):
    """Split-precision GEMM (operands as hi + lo 16-bit pairs, hi*hi + hi*lo + lo*hi on the 16-bit matrix cores) against an fp64
    product of the SAME fp32 operands: the error must be that of ~16 (bf16x3) / ~21 (f16x3) operand mantissa bits — 100x / 1000x
    below the plain 16-bit modes — through the skinny split-K family, the persistent 128x128 kernel and the 256x256 ping-pong
    kernel; EPI_STORE / EPI_GELU outputs are operand rows themselves (read back with unsplit_operand)."""
    from whisperseg_amd import _lib
    from whisperseg_amd.engine import DTYPES, SPLIT_BASE, split_operand, unsplit_operand
    base = SPLIT_BASE[dtype]
    g = torch.Generator(device="cuda").manual_seed(M * 31 + N * 7 + K + epi)
    Mp = (M + 255) // 256 * 256
    A = torch.rand(Mp, K, device="cuda", generator=g) * 2 - 1
    W = (torch.rand(N, K, device="cuda", generator=g) * 2 - 1) * K ** -0.5
    bias = torch.rand(N, device="cuda", generator=g) - 0.5
    res = torch.rand(Mp, N, device="cuda", generator=g) - 0.5
    As, Ws = split_operand(A, base), split_operand(W, base)
    assert torch.equal(unsplit_operand(As, base), (A.to(base).float() + (A - A.to(base).float()).to(base).float()))
    if epi == 2:
        out = torch.full((Mp, N), float("nan"), device="cuda")
    else:
        out = torch.full((Mp, 2 * N), -1, device="cuda", dtype=torch.int16)      # 0xffff words: NaN in both 16-bit types
    ws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    _lib.check(gpu_lib.wseg_debug_gemm(DTYPES[dtype][0], epi, M, N, K, As.data_ptr(), Ws.data_ptr(), bias.data_ptr(), res.data_ptr(),
                                       out.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr()))
    got_all = out if epi == 2 else unsplit_operand(out, base)
    assert torch.isnan(got_all[M:]).all() or M == Mp      # rows beyond M are never written
    worst, scale = 0.0, 1.0
    for lo in range(0, M, 8192):
        hi = min(M, lo + 8192)
        ref = A[lo:hi].double() @ W.double().T + bias.double()
        if epi == 1:
            ref = torch.nn.functional.gelu(ref)
        if epi == 2:
            ref = ref + res[lo:hi].double()
        got = got_all[lo:hi].double()
        assert torch.isfinite(got).all()
        worst = max(worst, (got - ref).abs().max().item())
        scale = max(scale, ref.abs().max().item())
    # bf16x3: operands to 2^-17 relative (hi + lo) plus the dropped lo*lo term, outputs re-split (2^-17): a few 1e-5 on O(1) sums.
    # f16x3: 2^-22 operands; what is left is fp32 accumulation order and the fast erf of the GELU epilogue (1.5e-7 absolute)
    tol = {"bf16x3": 6e-5, "f16x3": 6e-6}[dtype]
    assert worst <= tol * scale, (worst, scale)


M6_SHAPES = [sh for sh in X3_SHAPES if sh[2] % 64 == 0] + [(128000, 1280, 1280), (25000, 2560, 5120), (4096, 1280, 5120), (4096, 5120, 1280),
                                                            # the 256x256 kernel's K loop is peeled into leading edge / middle / trailing edge (hi, MX) tile
                                                            # pairs: one pair (first and last at once), two, five
                                                            (66000, 1280, 64), (8192, 2560, 128), (66000, 1280, 320),
                                                            # r05: M6-row outputs below the large-tile threshold through split-K copies of the 256x256 kernel + the
                                                            # M6-writing reduction (decoder fc1 at 112-270 slots); 480 rows: the >= 96-workgroup rule
                                                            (1024, 5120, 1280), (480, 5120, 1280), (480, 1280, 1280), (900, 1280, 5120)]


@pytest.mark.parametrize("M,N,K", M6_SHAPES)
@pytest.mark.parametrize("epi", [0, 1, 2])
def test_mixed_precision_gemm_matches_fp64(gpu_lib, M, N, K, epi):
    """WSEG_F16M6: hi*hi on the IEEE-half matrix cores + both cross terms on the block-scaled MX matrix cores (fp6 e2m3 codes with a
    power-of-two scale per 32 columns, v_mfma_scale_f32_16x16x128_f8f6f4), operands converted on the device by
    wseg_convert_operand (activation order / weight order), against an fp64 product of the same fp32 operands.  Error budget: the
    cross terms are 2^-11 of a product and carry 3 mantissa bits (2^-4 relative, block-scaled): ~2^-15.5 per product, random in
    sign — a few 1e-5 of the output scale (bf16x3 class, tools/precision_study.py "gemm=f16m6").  Every kernel family: skinny
    split-K (whole (hi, MX) tile pairs per split), 128x128 persistent, 256x256 ping-pong and its split-K form; run twice for
    bit-stability."""
    from whisperseg_amd import _lib
    from whisperseg_amd.engine import split_operand, unsplit_m6, unsplit_operand
    base = torch.float16
    g = torch.Generator(device="cuda").manual_seed(M * 31 + N * 7 + K + epi)
    Mp = (M + 255) // 256 * 256
    A = torch.rand(Mp, K, device="cuda", generator=g) * 2 - 1
    A[:, : K // 2] *= 0.02                                  # blocks of very different magnitude inside a row: the block scales matter
    A[::7, 5] = 37.5                                       # an outlier inside a block of small values
    W = (torch.rand(N, K, device="cuda", generator=g) * 2 - 1) * K ** -0.5
    bias = torch.rand(N, device="cuda", generator=g) - 0.5
    res = torch.rand(Mp, N, device="cuda", generator=g) - 0.5
    As, Ws = split_operand(A, base), split_operand(W, base)
    Am, Wm = torch.empty_like(As), torch.empty_like(Ws)
    _lib.check(gpu_lib.wseg_convert_operand(As.data_ptr(), Am.data_ptr(), Mp, K, 0, _lib.stream_ptr()))
    _lib.check(gpu_lib.wseg_convert_operand(Ws.data_ptr(), Wm.data_ptr(), N, K, 1, _lib.stream_ptr()))
    # the hi halves travel unchanged: 64 hi words of every 128-word block
    assert torch.equal(Am.view(Mp, K // 64, 128)[:, :, :64].reshape(Mp, K // 32, 32), As.view(Mp, K // 32, 2, 32)[:, :, 0])
    ws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    outs = []
    for rep in range(2):
        if epi == 2:
            out = torch.full((Mp, N), float("nan"), device="cuda")
        else:
            out = torch.full((Mp, 2 * N), -1, device="cuda", dtype=torch.int16)
        _lib.check(gpu_lib.wseg_debug_gemm(5, epi, M, N, K, Am.data_ptr(), Wm.data_ptr(), bias.data_ptr(), res.data_ptr(),
                                           out.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr()))
        torch.cuda.synchronize()
        outs.append(out)
    assert torch.equal(outs[0][:M].view(torch.int32), outs[1][:M].view(torch.int32))
    # the conversion itself: hi + fp6(lo) of the M6 rows reproduces the operand to ~2^-15 of each 32-column block's maximum
    blockmax = A.abs().view(Mp, K // 32, 32).amax(-1, keepdim=True).expand(-1, -1, 32).reshape(Mp, K)
    assert ((unsplit_m6(Am) - A).abs() <= 2.0 ** -14 * blockmax + 1e-30).all()
    # EPI_STORE / EPI_GELU outputs are the next GEMM's operand: M6 rows from the LDS-staged epilogues of the large-tile kernels
    # (N % 64 == 0 there), hi | lo IEEE-half rows from the skinny family
    if epi == 2:
        got_all = outs[0]
    elif gpu_lib.wseg_debug_gemm_out_is_mx(5, M, N, K):
        got_all = torch.full((Mp, N), float("nan"), device="cuda")
        got_all[:M] = unsplit_m6(outs[0][:M])
    else:
        got_all = unsplit_operand(outs[0], base)
    assert torch.isnan(got_all[M:]).all() or M == Mp
    worst, scale = 0.0, 1.0
    for lo in range(0, M, 8192):
        hi = min(M, lo + 8192)
        ref = A[lo:hi].double() @ W.double().T + bias.double()
        if epi == 1:
            ref = torch.nn.functional.gelu(ref)
        if epi == 2:
            ref = ref + res[lo:hi].double()
        got = got_all[lo:hi].double()
        assert torch.isfinite(got).all()
        worst = max(worst, (got - ref).abs().max().item())
        scale = max(scale, ref.abs().max().item())
    assert worst <= 1e-4 * scale, (worst, scale)


# Shapes that reach the 256x256 ping-pong kernel (N % 256 == 0, >= 192 tiles): fewer tiles than workgroups x 2,
# several tiles per workgroup (the K-tile stream crosses output tiles and holds the prefetch back over the epilogue),
# the minimum K (2 K tiles), an odd number of K tiles, M not a tile multiple, and the bench's own row count.
PP_SHAPES = [(12800, 1280, 1280), (8192, 2560, 128), (10000, 3840, 192), (66000, 1280, 320), (128000, 1280, 1280),
             (30000, 5120, 1280), (25000, 2560, 5120)]


@pytest.mark.parametrize("M,N,K", PP_SHAPES)
@pytest.mark.parametrize("epi", [0, 1, 2])
def test_pingpong_gemm_matches_torch(gpu_lib, M, N, K, epi):
    from whisperseg_amd import _lib
    g = torch.Generator(device="cuda").manual_seed(M + N * 3 + K * 5 + epi)
    Mp = (M + 255) // 256 * 256
    A = (torch.rand(Mp, K, device="cuda", generator=g) * 2 - 1).to(torch.bfloat16)
    W = ((torch.rand(N, K, device="cuda", generator=g) * 2 - 1) * K ** -0.5).to(torch.bfloat16)
    bias = (torch.rand(N, device="cuda", generator=g) - 0.5).to(torch.bfloat16)
    od = torch.float32 if epi == 2 else torch.bfloat16        # the residual-stream epilogue reads / writes fp32
    res = (torch.rand(Mp, N, device="cuda", generator=g) - 0.5).to(od)
    ws = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
    outs = []
    for rep in range(3):          # race screen: a mis-ordered LDS-DMA / ds_read shows up as run-to-run differences
        out = torch.full((Mp, N), float("nan"), device="cuda", dtype=od)
        _lib.check(gpu_lib.wseg_debug_gemm(1, epi, M, N, K, A.data_ptr(), W.data_ptr(), bias.data_ptr(), res.data_ptr(),
                                           out.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr()))
        torch.cuda.synchronize()
        outs.append(out)
    bits = torch.int32 if epi == 2 else torch.int16
    assert torch.equal(outs[0][:M].view(bits), outs[1][:M].view(bits))
    assert torch.equal(outs[0][:M].view(bits), outs[2][:M].view(bits))
    assert torch.isnan(outs[0][M:]).all()                      # rows past M are never written
    worst = 0.0
    for lo in range(0, M, 16384):                              # reference in row blocks (fp32 128000 x 5120 would be 2.6 GB)
        hi = min(M, lo + 16384)
        ref = A[lo:hi].float() @ W.float().T + bias.float()
        if epi == 1:
            ref = torch.nn.functional.gelu(ref)
        if epi == 2:
            ref = ref + res[lo:hi].float()
        worst = max(worst, (outs[0][lo:hi].float() - ref).abs().max().item())
    assert worst <= 2e-2, worst                                # O(1) outputs rounded to bf16 (2^-9 relative)


F32_BIG_SCRIPT = r"""
import sys, torch
sys.path.insert(0, sys.argv[2])
from whisperseg_amd import _lib
lib = _lib.load(require_device=True)
outs = []
for (M, N, K, epi) in ((8192, 1280, 1280, 0), (8100, 1280, 5120, 2), (33000, 640, 256, 1), (200, 1280, 1280, 2), (7, 3840, 1280, 0), (1000, 1280, 5120, 1)):
    g = torch.Generator(device="cuda").manual_seed(M + epi)
    mp = (M + 255) // 256 * 256
    A = torch.rand(mp, K, device="cuda", generator=g) * 2 - 1
    W = torch.rand(N, K, device="cuda", generator=g) * 2 - 1
    b = torch.rand(N, device="cuda", generator=g)
    r = torch.rand(mp, N, device="cuda", generator=g)
    o = torch.zeros(mp, N, device="cuda")
    _lib.check(lib.wseg_debug_gemm(0, epi, M, N, K, A.data_ptr(), W.data_ptr(), b.data_ptr(), r.data_ptr(), o.data_ptr(), None, 0,
                                   _lib.stream_ptr()))
    ref = A[:M].double() @ W.double().T + b.double()
    if epi == 1: ref = torch.nn.functional.gelu(ref)
    if epi == 2: ref = ref + r[:M].double()
    assert (o[:M].double() - ref).abs().max().item() < 2e-3, (M, N, K, epi)
    assert float(o[M:].abs().max()) == 0.0 if mp > M else True          # rows past M are not written
    outs.append(o[:M].cpu())
torch.save(outs, sys.argv[1])
"""


def test_f32_gemm_kernels_are_bit_identical(gpu_lib, tmp_path):
    """The exact-parity mode's GEMM runs on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: 128x128 tiles for large problems,
    64x64 otherwise, 32x32 tiles of v_mfma_f32_16x16x4_f32 when those would not fill the chip).  That instruction is a k-ordered fmaf chain bit for bit, so the MFMA kernels must reproduce the two VALU
    kernels (64x64 tile / 4x4 per thread; 128x128 tile / 8x8 per thread with packed FMAs) exactly.  WSEG_F32_GEMM selects the
    kernel; the knob is read once per process, hence one child process per kernel."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "f32gemm.py"
    script.write_text(F32_BIG_SCRIPT)
    outs = {}
    for mode in ("mfma", "mfma64", "valu128", "valu64"):
        out = tmp_path / f"{mode}.pt"
        subprocess.check_call([sys.executable, str(script), str(out), root], env={**os.environ, "WSEG_F32_GEMM": mode})
        outs[mode] = torch.load(out)
    for mode in ("mfma64", "valu128", "valu64"):
        for a, b in zip(outs["mfma"], outs[mode]):
            assert torch.equal(a, b), mode


def test_f16x3_gemm_operands_saturate(gpu_lib):
    """A GELU output (the fc2 operand) beyond the fp16 range saturates at 65504 in the f16x3 mode instead of poisoning the row
    with inf - inf: epilogue 1 (GELU) writes operand rows, which are read back here."""
    from whisperseg_amd import _lib
    from whisperseg_amd.engine import DTYPES, split_operand, unsplit_operand
    M, N, K = 256, 128, 64
    A = torch.zeros(M, K, device="cuda"); A[:, 0] = 300.0
    W = torch.zeros(N, K, device="cuda"); W[:, 0] = 300.0            # products of 90 000 > 65 504
    W[1, 0] = 1.0
    bias = torch.zeros(N, device="cuda")
    out = torch.zeros((M, 2 * N), device="cuda", dtype=torch.int16)
    ws = torch.empty(16 << 20, dtype=torch.uint8, device="cuda")
    _lib.check(gpu_lib.wseg_debug_gemm(DTYPES["f16x3"][0], 1, M, N, K, split_operand(A, torch.float16).data_ptr(),
                                       split_operand(W, torch.float16).data_ptr(), bias.data_ptr(), None, out.data_ptr(), ws.data_ptr(),
                                       ws.numel(), _lib.stream_ptr()))
    got = unsplit_operand(out, torch.float16)
    assert torch.isfinite(got).all()
    assert float(got[0, 0]) == 65504.0 and abs(float(got[0, 1]) - 300.0) < 1e-2


RESID_LN_SHAPES = [  # M, N, K: skinny split-K + fused reduction (<= 1024 rows), 256x256 ping-pong split-K (long K, 2048+ rows:
    # uneven K-tile shares 27/27/26, 13/13/.., ragged last row tile), un-split GEMM + separate LayerNorm
    (32, 1280, 1280), (480, 1280, 5120), (1024, 1280, 5120), (2048, 1280, 5120), (4096, 1280, 5120), (4000, 1280, 5120),
    (3072, 1280, 5120), (4096, 1280, 1280), (2100, 768, 3072), (5120, 1280, 5120),
]


@pytest.mark.parametrize("M,N,K", RESID_LN_SHAPES)
@pytest.mark.parametrize("dtype", ["bf16", "f16", "f16x3", "f16m6", "f32"])
def test_gemm_resid_layernorm_step(gpu_lib, M, N, K, dtype):
    """The decoder's fused step x += A W^T + b; y = LayerNorm(x) through every kernel family the row count selects (f16m6, the
    default of r04-r05: operands and the LayerNorm output are M6 rows)."""
    from whisperseg_amd import _lib
    from whisperseg_amd.engine import DTYPES, SPLIT_BASE, split_operand, unsplit_m6, unsplit_operand
    if dtype == "f32" and M * N * K > 3e9:
        pytest.skip("f32 exact kernel is for small problems")
    x3 = dtype in SPLIT_BASE
    m6 = dtype == "f16m6"
    td = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32, "f16x3": torch.float32, "f16m6": torch.float32}[dtype]
    g = torch.Generator(device="cuda").manual_seed(M * 13 + N * 5 + K)
    Mp = (M + 255) // 256 * 256
    A = (torch.rand(Mp, K, device="cuda", generator=g) * 2 - 1).to(td)
    W = ((torch.rand(N, K, device="cuda", generator=g) * 2 - 1) * K ** -0.5).to(td)
    bias = (torch.rand(N, device="cuda", generator=g) - 0.5).to(td)
    gam = (torch.rand(N, device="cuda", generator=g) + 0.5).to(td)
    bet = (torch.rand(N, device="cuda", generator=g) - 0.5).to(td)
    x0 = (torch.rand(Mp, N, device="cuda", generator=g) - 0.5)
    x = x0.clone()
    Ao, Wo = (split_operand(A, SPLIT_BASE[dtype]), split_operand(W, SPLIT_BASE[dtype])) if x3 else (A, W)
    if m6:
        As, Ws = Ao, Wo
        Ao, Wo = torch.empty_like(As), torch.empty_like(Ws)
        _lib.check(gpu_lib.wseg_convert_operand(As.data_ptr(), Ao.data_ptr(), Mp, K, 0, _lib.stream_ptr()))
        _lib.check(gpu_lib.wseg_convert_operand(Ws.data_ptr(), Wo.data_ptr(), N, K, 1, _lib.stream_ptr()))
    y = torch.full((Mp, 2 * N if x3 else N), float("nan"), device="cuda", dtype=td) if not x3 else \
        torch.full((Mp, 2 * N), 0x7e00, device="cuda", dtype=torch.int16)        # NaN halves
    ws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    _lib.check(gpu_lib.wseg_debug_gemm_resid_ln(DTYPES[dtype][0], M, N, K, Ao.data_ptr(), Wo.data_ptr(), bias.data_ptr(), x.data_ptr(),
                                                gam.data_ptr(), bet.data_ptr(), y.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr()))
    xr = x0[:M].double() + A[:M].double() @ W.double().T + bias.double()
    yr = torch.nn.functional.layer_norm(xr, (N,), gam.double(), bet.double(), 1e-5)
    tol = {"bf16": 2e-2, "f16": 3e-3, "f32": 2e-5, "f16x3": 2e-5, "f16m6": 1e-4}[dtype]
    # x is the fp32 residual stream in every mode: products of exactly representable operands, fp32 accumulation (f16m6: the cross
    # terms carry 3 mantissa bits, ~2^-15.5 per product — the bound of test_mixed_precision_gemm_matches_fp64)
    assert (x[:M].double() - xr).abs().max().item() <= (1e-4 if m6 else 2e-5) * max(1.0, xr.abs().max().item())
    got = (unsplit_m6(y[:M]) if m6 else unsplit_operand(y[:M], SPLIT_BASE[dtype]) if x3 else y[:M]).double()
    assert torch.isfinite(got).all()
    assert (got - yr).abs().max().item() <= tol * max(1.0, yr.abs().max().item())
    assert torch.equal(x[M:], x0[M:])                                    # rows beyond M untouched


def test_lane_exchanges(gpu_lib):
    """lane_xor<M> (DPP / v_permlane swaps; every wave_sum / wave_max / attention lane reduction of the library) is the exact
    exchange l <-> l ^ M that __shfl_xor was."""
    from whisperseg_amd import _lib
    out = torch.zeros(6 * 64, dtype=torch.int32, device="cuda")
    _lib.check(gpu_lib.wseg_debug_lane_xor(out.data_ptr(), _lib.stream_ptr()))
    got = out.cpu().view(6, 64)
    lanes = torch.arange(64)
    for m in range(6):
        assert torch.equal(got[m], ((lanes ^ (1 << m)) * 7 + 3).to(torch.int32)), 1 << m

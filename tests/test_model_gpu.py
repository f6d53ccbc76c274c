"""HIP encoder / decoder / beam search (through the C-ABI) vs the oracle on seeded random weights.

f32 mode  : exact-parity mode of the same kernels; tolerance 5e-4 abs on O(1) activations / logits and
            token-exact sequences.
bf16 mode : production mode; weights are rounded to bf16 on BOTH sides, tolerance is relative to the
            activation scale (bf16 has 8 mantissa bits; ~1e-2 after 2-4 layers is the expected class).
"""
import numpy as np
import pytest
import torch

from oracle import whisper_ref as R

pytestmark = pytest.mark.gpu

PROMPT = [1100, 1102, 1103]
EOS = 1101


def hf_cfg(d=128, heads=2, layers=2, ffn=512, vocab=1280):
    return dict(d_model=d, encoder_attention_heads=heads, decoder_attention_heads=heads, encoder_layers=layers,
                decoder_layers=layers, encoder_ffn_dim=ffn, decoder_ffn_dim=ffn, vocab_size=vocab, num_mel_bins=80,
                max_source_positions=500, max_target_positions=448)


def make(cfg, dtype, seed=1):
    from whisperseg_amd.engine import Engine
    rc = R.RefConfig.from_hf_dict(cfg)
    sd = R.random_state_dict(rc, seed=seed)
    if dtype in ("bf16", "f16"):      # bf16-representable weights are exact in f16 too
        sd = {k: v.to(torch.bfloat16).float() for k, v in sd.items()}
    eng = Engine.from_state_dict(sd, cfg, "cuda:0", dtype)
    return rc, sd, eng


def feats(n, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n, 80, 1000, generator=g) * 0.5


@pytest.mark.parametrize("dtype,tol", [("f32", 5e-4), ("bf16", 6e-2), ("f16", 1e-2), ("bf16x3", 5e-4), ("f16x3", 5e-4), ("f16m6", 5e-4)])
@pytest.mark.parametrize("n", [1, 3])
def test_encoder_matches_oracle(gpu_lib, dtype, tol, n):
    cfg = hf_cfg()
    rc, sd, eng = make(cfg, dtype)
    x = feats(n)
    want = R.encoder_forward(sd, rc, x)
    got = eng.encode(x.cuda()).float().cpu()
    assert got.shape == want.shape
    err = (got - want).abs().max().item()
    assert err <= tol * max(1.0, want.abs().max().item()), err


@pytest.mark.parametrize("dtype,tol", [("f32", 5e-4), ("bf16", 8e-2), ("f16", 1.5e-2), ("bf16x3", 5e-4), ("f16x3", 5e-4), ("f16m6", 5e-4)])
def test_encoder_wider_geometry(gpu_lib, dtype, tol):
    """d=256 / 4 heads / ffn 1024: exercises multi-tile N and more than two heads."""
    cfg = hf_cfg(d=256, heads=4, layers=2, ffn=1024)
    rc, sd, eng = make(cfg, dtype, seed=3)
    x = feats(2, seed=5)
    want = R.encoder_forward(sd, rc, x)
    got = eng.encode(x.cuda()).float().cpu()
    err = (got - want).abs().max().item()
    assert err <= tol * max(1.0, want.abs().max().item()), err


def gen_params(nb, ml):
    return R.GenParams(prompt=PROMPT, eos_token_id=EOS, pad_token_id=EOS, max_length=ml, num_beams=nb,
                       suppress_tokens=[5, 6, 7, 200], begin_suppress_tokens=[220, EOS])


@pytest.mark.parametrize("dtype", ["f32", "bf16x3", "f16x3", "f16m6"])      # the split-precision modes meet the exact mode's bound
@pytest.mark.parametrize("nb", [1, 4])
def test_first_logits_f32(gpu_lib, nb, dtype):
    cfg = hf_cfg()
    rc, sd, eng = make(cfg, dtype)
    x = feats(2)
    gp = gen_params(nb, 8)
    _, want = R.generate(sd, rc, x, gp, return_first_logits=True)
    _, _, got = eng.generate(x.cuda(), PROMPT, EOS, EOS, max_length=8, num_beams=nb, suppress_tokens=gp.suppress_tokens,
                             begin_suppress_tokens=gp.begin_suppress_tokens, return_first_logits=True)
    err = (got.cpu() - want).abs().max().item()
    assert err <= 1e-3, err


@pytest.mark.parametrize("dtype", ["f32", "bf16x3", "f16x3", "f16m6"])
@pytest.mark.parametrize("nb,ml", [(1, 12), (1, 40), (4, 12), (4, 40), (2, 20)])
def test_generate_tokens_f32_random_weights(gpu_lib, nb, ml, dtype):
    cfg = hf_cfg()
    rc, sd, eng = make(cfg, dtype)
    x = feats(3)
    gp = gen_params(nb, ml)
    want = R.generate(sd, rc, x, gp)
    toks, lens = eng.generate(x.cuda(), PROMPT, EOS, EOS, max_length=ml, num_beams=nb, suppress_tokens=gp.suppress_tokens,
                              begin_suppress_tokens=gp.begin_suppress_tokens)
    toks, lens = toks.cpu().numpy(), lens.cpu().numpy()
    for i in range(3):
        a = R.canonical(want[i].tolist(), 3, EOS, PROMPT)
        b = R.canonical(toks[i, :lens[i]].tolist(), 3, EOS, PROMPT)
        assert a == b, (i, a, b)


@pytest.mark.parametrize("dtype", ["f32", "f16x3", "f16m6"])
@pytest.mark.parametrize("nb", [5, 8])
def test_five_to_eight_beams(gpu_lib, nb, dtype):
    """num_beams 5..8 (MAX_BEAMS = 8): the 24-bit cross-K/V kernel exists for up to 4 beams, above that cross-attention runs the
    general fp32-K/V kernel — in f16m6 (the default of r04-r05) its hi | lo output rows are converted to M6 rows for the co-proj GEMM
    (the default mode used to reject these calls after the encoder had already run).  Token-exact vs the oracle."""
    cfg = hf_cfg()
    rc, sd, eng = make(cfg, dtype)
    x = feats(3)
    gp = gen_params(nb, 16)
    want, want_logits = R.generate(sd, rc, x, gp, return_first_logits=True)
    toks, lens, got_logits = eng.generate(x.cuda(), PROMPT, EOS, EOS, max_length=16, num_beams=nb, suppress_tokens=gp.suppress_tokens,
                                          begin_suppress_tokens=gp.begin_suppress_tokens, return_first_logits=True)
    # first-step logits of EVERY beam row (r05: beams 4..7 of the general cross-attention kernel read their split-K queries from
    # uninitialised LDS in every mode but f32 — errors of 0.1-3 on the logits while the tokens of flat random weights still agreed)
    assert (got_logits.cpu() - want_logits).abs().max().item() <= 1e-3
    toks, lens = toks.cpu().numpy(), lens.cpu().numpy()
    for i in range(3):
        a = R.canonical(want[i].tolist(), 3, EOS, PROMPT)
        b = R.canonical(toks[i, :lens[i]].tolist(), 3, EOS, PROMPT)
        if a != b:      # 8 hypotheses over flat random-weight logits: a differing result must be a near-tie under the ORACLE's own scores
            from test_large_geometry_gpu import assert_equally_scored
            assert dtype != "f32", (i, a, b)
            assert_equally_scored(sd, rc, x[i:i + 1], gp, toks[i, :lens[i]].tolist(), want[i].tolist(), ("beams", nb, dtype, i))


def test_real_vocab_logits_bf16(gpu_lib):
    """Whisper's real vocabulary size (51865, not a tile multiple) through the tied LM head."""
    cfg = hf_cfg(vocab=51865)
    rc, sd, eng = make(cfg, "bf16", seed=7)
    x = feats(2, seed=9)
    prompt, eos = [50258, 50259, 50363], 50257
    gp = R.GenParams(prompt=prompt, eos_token_id=eos, pad_token_id=eos, max_length=6, num_beams=4,
                     suppress_tokens=[1, 2, 7, 50258], begin_suppress_tokens=[220, eos])
    _, want = R.generate(sd, rc, x, gp, return_first_logits=True)
    _, _, got = eng.generate(x.cuda(), prompt, eos, eos, max_length=6, num_beams=4, suppress_tokens=gp.suppress_tokens,
                             begin_suppress_tokens=gp.begin_suppress_tokens, return_first_logits=True)
    got = got.cpu()
    assert got.shape == want.shape == (8, 51865)
    scale = want.abs().max().item()
    assert (got - want).abs().max().item() <= 8e-2 * max(scale, 1.0)
    # beams of one window are identical at the first step
    assert torch.equal(got[0], got[1])


def test_base_geometry_bf16(gpu_lib):
    """BASELINE config[1] geometry (whisperseg-base: d 512, 8 heads, 6+6 layers, ffn 2048, vocab 51865), seeded random
    weights, bf16: encoder output and first-step logits vs the oracle; beams of a window agree at the first step."""
    cfg = hf_cfg(d=512, heads=8, layers=6, ffn=2048, vocab=51865)
    rc, sd, eng = make(cfg, "bf16", seed=11)
    x = feats(2, seed=13)
    want_enc = R.encoder_forward(sd, rc, x)
    got_enc = eng.encode(x.cuda()).float().cpu()
    assert (got_enc - want_enc).abs().max().item() <= 0.1 * max(1.0, want_enc.abs().max().item())
    prompt, eos = [50258, 50259, 50363], 50257
    gp = R.GenParams(prompt=prompt, eos_token_id=eos, pad_token_id=eos, max_length=5, num_beams=4)
    _, want = R.generate(sd, rc, x, gp, return_first_logits=True)
    toks, lens, got = eng.generate(x.cuda(), prompt, eos, eos, max_length=5, num_beams=4, return_first_logits=True)
    got = got.cpu()
    assert (got - want).abs().max().item() <= 0.1 * max(1.0, want.abs().max().item())
    # cosine similarity of the logit rows: a transposed / mis-indexed head would destroy it
    cos = torch.nn.functional.cosine_similarity(got, want, dim=1)
    assert cos.min().item() > 0.999
    assert lens.tolist() == [5, 5]


@pytest.mark.parametrize("mode", ["f16x3", "f16m6"])
def test_base_geometry_32_windows_default_mode(gpu_lib, mode):
    """BASELINE configs[1] as the product runs it: whisperseg-base geometry x 32 windows x 4 beams in the default mode f16x3 (and in f16m6,
    the default of r04-r05).
    The oracle (torch-CPU fp32) is run on a subset of the windows (windows are independent): encoder output and first-step
    logits within the split modes' bound, beam sequences token-exact; the other windows through the size-independent
    properties (beams equal at step 1, the same window gives the same tokens wherever it sits in the batch)."""
    cfg = hf_cfg(d=512, heads=8, layers=6, ffn=2048, vocab=51865)
    rc, sd, eng = make(cfg, mode, seed=11)
    x = feats(32, seed=17)
    sub = [0, 13, 31]
    prompt, eos = [50258, 50259, 50363], 50257
    gp = R.GenParams(prompt=prompt, eos_token_id=eos, pad_token_id=eos, max_length=8, num_beams=4)
    want_enc = R.encoder_forward(sd, rc, x[sub])
    got_enc = eng.encode(x.cuda()).float().cpu()
    assert (got_enc[sub] - want_enc).abs().max().item() <= 1e-3 * max(1.0, want_enc.abs().max().item())
    want_seq, want = R.generate(sd, rc, x[sub], gp, return_first_logits=True)
    toks, lens, got = eng.generate(x.cuda(), prompt, eos, eos, max_length=8, num_beams=4, return_first_logits=True)
    got, toks, lens = got.cpu(), toks.cpu(), lens.cpu()
    rows = [4 * i + b for i in sub for b in range(4)]
    scale = max(1.0, want.abs().max().item())
    assert (got[rows] - want).abs().max().item() <= 2e-3 * scale
    assert torch.nn.functional.cosine_similarity(got[rows], want, dim=1).min().item() > 0.999999
    assert torch.equal(got[0::4], got[1::4]) and torch.equal(got[0::4], got[3::4])
    for k, i in enumerate(sub):
        a = R.canonical(want_seq[k].tolist(), 3, eos, prompt)
        b = R.canonical(toks[i, :lens[i]].tolist(), 3, eos, prompt)
        if a != b:      # flat random-weight logits: a differing beam result must be a near-tie under the oracle's own scores
            from test_large_geometry_gpu import assert_equally_scored
            assert_equally_scored(sd, rc, x[i:i + 1], gp, toks[i, :lens[i]].tolist(), want_seq[k].tolist(), ("base32", i))
    perm = torch.randperm(32, generator=torch.Generator().manual_seed(2))
    toks_p, lens_p = eng.generate(x[perm].cuda(), prompt, eos, eos, max_length=8, num_beams=4)
    assert torch.equal(toks_p.cpu(), toks[perm]) and torch.equal(lens_p.cpu(), lens[perm])


@pytest.mark.parametrize("dtype,tol_enc,tol_logit,cos_min", [("bf16", 8e-2, 0.1, 0.999), ("bf16x3", 1e-3, 2e-3, 0.999999),
                                                             ("f16x3", 1e-3, 2e-3, 0.999999), ("f16m6", 1e-3, 2e-3, 0.999999)])
def test_pingpong_gemm_epilogues_in_the_model_bf16(gpu_lib, dtype, tol_enc, tol_logit, cos_min):
    """50 windows at d=512 / 8 heads / ffn 2048 (2+2 layers): every large GEMM of the path has >= 192 tiles of 256x256, so
    conv2 (+pos-emb), the QKV head split incl. V^T, o-proj / fc2 (residual), fc1 (GELU) and the cross-K/V head split all
    run in the ping-pong kernel — encoder output and first-step logits vs the oracle, plus the size-independent
    properties: beams of a window agree at the first step, and a permutation of the windows permutes the results."""
    cfg = hf_cfg(d=512, heads=8, layers=2, ffn=2048, vocab=1280)
    rc, sd, eng = make(cfg, dtype, seed=21)
    x = feats(50, seed=23)
    want_enc = R.encoder_forward(sd, rc, x)
    got_enc = eng.encode(x.cuda()).float().cpu()
    assert (got_enc - want_enc).abs().max().item() <= tol_enc * max(1.0, want_enc.abs().max().item())
    gp = gen_params(4, 6)
    _, want = R.generate(sd, rc, x, gp, return_first_logits=True)
    toks, lens, got = eng.generate(x.cuda(), PROMPT, EOS, EOS, max_length=6, num_beams=4, suppress_tokens=gp.suppress_tokens,
                                   begin_suppress_tokens=gp.begin_suppress_tokens, return_first_logits=True)
    got = got.cpu()
    assert (got - want).abs().max().item() <= tol_logit * max(1.0, want.abs().max().item())
    assert torch.nn.functional.cosine_similarity(got, want, dim=1).min().item() > cos_min
    assert torch.equal(got[0::4], got[1::4])
    perm = torch.randperm(50, generator=torch.Generator().manual_seed(1))
    toks_p, lens_p, got_p = eng.generate(x[perm].cuda(), PROMPT, EOS, EOS, max_length=6, num_beams=4,
                                         suppress_tokens=gp.suppress_tokens, begin_suppress_tokens=gp.begin_suppress_tokens,
                                         return_first_logits=True)
    assert torch.equal(got_p.cpu()[0::4], got[0::4][perm])          # bit-identical rows: no dependence on the batch slot
    assert torch.equal(toks_p.cpu(), toks.cpu()[perm]) and torch.equal(lens_p.cpu(), lens.cpu()[perm])


def test_packed_cross_attention_is_bit_equal_to_the_fma_kernel(gpu_lib):
    """dec_cross_attn_pk_kernel (v_pk_fma_f32 over beam pairs, DPP row sums) against dec_cross_attn_kernel (WSEG_CROSS_NO_PK=1,
    a process-wide switch read once: two child processes): first-step logits and tokens identical for 1, 2, 3 and 4 beams."""
    import os, subprocess, sys, tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, torch
sys.path.insert(0, sys.argv[2])
from whisperseg_amd.engine import Engine
cfg = dict(d_model=512, encoder_attention_heads=8, decoder_attention_heads=8, encoder_layers=1, decoder_layers=3,
           encoder_ffn_dim=1024, decoder_ffn_dim=1024, vocab_size=1280, num_mel_bins=80, max_source_positions=500, max_target_positions=448)
eng = Engine.random(cfg, "cuda:0", "bf16", seed=3)
x = torch.randn(10, 80, 1000, generator=torch.Generator().manual_seed(5)) * 0.5
out = {}
for nb in (1, 2, 3, 4):
    t, l, fl = eng.generate(x.cuda(), [1100, 1102, 1103], 1101, 1101, max_length=14, num_beams=nb, return_first_logits=True)
    out[nb] = (t.cpu(), l.cpu(), fl.cpu())
torch.save(out, sys.argv[1])
'''
    with tempfile.TemporaryDirectory() as tmp:
        res = {}
        for tag, env in (("pk", {}), ("fma", {"WSEG_CROSS_NO_PK": "1"})):
            path = os.path.join(tmp, tag + ".pt")
            subprocess.check_call([sys.executable, "-c", code, path, root], env={**os.environ, **env})
            res[tag] = torch.load(path)
    for nb in (1, 2, 3, 4):
        for got, want in zip(res["pk"][nb], res["fma"][nb]):
            assert torch.equal(got, want), nb


F32_ATTN_SCRIPT = r"""
import sys, torch
sys.path.insert(0, sys.argv[2])
from whisperseg_amd.engine import Engine
cfg = dict(d_model=256, encoder_attention_heads=4, decoder_attention_heads=4, encoder_layers=2, decoder_layers=1, encoder_ffn_dim=512,
           decoder_ffn_dim=512, vocab_size=1280, num_mel_bins=80, max_source_positions=500, max_target_positions=448)
eng = Engine.random(cfg, "cuda:0", "f32", seed=5)
x = torch.randn(6, 80, 1000, device="cuda", generator=torch.Generator(device="cuda").manual_seed(9)) * 0.7
torch.save(eng.encode(x).cpu(), sys.argv[1])
"""


def test_f32_encoder_attention_kernels_are_bit_identical(gpu_lib, tmp_path):
    """Exact-parity mode: the encoder attention on the fp32 matrix cores (k-ordered fmaf chains: scores over the head
    dimension, probability sum and P·V over ascending keys, the row maximum from a first pass) against the one-thread-per-query
    kernel it restates — the encoder outputs must be equal bit for bit (WSEG_F32_ATTN=naive selects the old kernel)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "f32attn.py"
    script.write_text(F32_ATTN_SCRIPT)
    outs = []
    for env in ({}, {"WSEG_F32_ATTN": "naive"}):
        out = tmp_path / f"o{len(outs)}.pt"
        subprocess.check_call([sys.executable, str(script), str(out), root], env={**os.environ, **env})
        outs.append(torch.load(out))
    assert torch.isfinite(outs[0]).all()
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("positions", [251, 384])
@pytest.mark.parametrize("dtype,tol", [("f32", 5e-4), ("f16x3", 5e-4), ("f16m6", 5e-4), ("bf16", 6e-2)])
def test_other_window_lengths(gpu_lib, dtype, tol, positions):
    """Encoder positions other than 500 (r06: the V^T of the encoder attention is written in MFMA operand order by the q | k | v epilogue —
    4 consecutive positions per 8-byte store when the window length is a multiple of 4, element by element otherwise — and read by LDS-DMA
    in 64-key tiles): 251 positions (odd: the element-wise writer, a ragged last key tile, windows starting at odd GEMM rows) and 384 (a
    whole number of key tiles), 3 windows, encoder output and greedy tokens against the oracle."""
    cfg = dict(hf_cfg(), max_source_positions=positions)
    rc, sd, eng = make(cfg, dtype, seed=23)
    g = torch.Generator().manual_seed(positions)
    x = torch.randn(3, 80, 2 * positions, generator=g) * 0.5
    want = R.encoder_forward(sd, rc, x)
    got = eng.encode(x.cuda()).float().cpu()
    assert got.shape == want.shape
    assert (got - want).abs().max().item() <= tol * max(1.0, want.abs().max().item())
    if dtype != "bf16":
        gp = R.GenParams(prompt=PROMPT, eos_token_id=EOS, pad_token_id=EOS, max_length=10, num_beams=1)
        want_t = R.generate(sd, rc, x, gp)
        toks, lens = eng.generate(x.cuda(), PROMPT, EOS, EOS, max_length=10, num_beams=1)
        toks, lens = toks.cpu(), lens.cpu()
        for i in range(3):
            assert R.canonical(toks[i, :lens[i]].tolist(), 3, EOS, PROMPT) == R.canonical(want_t[i].tolist(), 3, EOS, PROMPT), (dtype, positions, i)

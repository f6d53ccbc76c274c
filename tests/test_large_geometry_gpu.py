"""whisperseg-large geometry (d 1280, 20 heads, ffn 5120, vocab 51865) on the GPU — the configuration bench.py times.

(a) 2 + 2 layers (cheap enough for the CPU oracle): encoder output, first-step logits and tokens vs oracle/whisper_ref.py at
    8 windows (skinny / split-K decode plans, 128x128 encoder tiles) and at 256 windows x 4 beams = 1024 decode rows (the
    bench's row count: 256x256 ping-pong encoder tiles, the 1024-row decode plans, the 51 968-wide LM head) — windows are
    independent, so the oracle is run on a subset of the 256 windows.
(b) the full 32 + 32 layers at 8 and 120 windows through size-independent properties: determinism, window-permutation
    equivariance, beams equal at the first step, bf16 first-step logits vs the exact-parity f32 mode of the same kernels; the split
    modes at 16 windows against the f32 mode: logits, and whole beam sequences equal or oracle-scored near-ties.
Tolerances: f32 mode 1e-3 abs on logits and token-exact; bf16 cosine >= 0.999 per logit row and 10 % of the logit scale
(bf16 has 8 mantissa bits; same class as tests/test_model_gpu.py)."""
import numpy as np
import pytest
import torch

from oracle import whisper_ref as R

pytestmark = pytest.mark.gpu

PROMPT, EOS = [50258, 50259, 50363], 50257
SUP, BSUP = [1, 2, 7, 50258], [220, EOS]


def large_cfg(layers):
    return dict(d_model=1280, encoder_attention_heads=20, decoder_attention_heads=20, encoder_layers=layers,
                decoder_layers=layers, encoder_ffn_dim=5120, decoder_ffn_dim=5120, vocab_size=51865, num_mel_bins=80,
                max_source_positions=500, max_target_positions=448)


def feats(n, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n, 80, 1000, generator=g) * 0.5


@pytest.fixture(scope="module")
def two_layer():
    """Seeded weights, rounded to bf16 on BOTH sides so that the oracle and both engine modes see identical parameters."""
    from whisperseg_amd.engine import Engine
    cfg = large_cfg(2)
    rc = R.RefConfig.from_hf_dict(cfg)
    sd = {k: v.to(torch.bfloat16).float() for k, v in R.random_state_dict(rc, seed=31, fast=True).items()}
    engines = {dt: Engine.from_state_dict(sd, cfg, "cuda:0", dt) for dt in ("f32", "bf16", "f16", "f16x3", "bf16x3", "f16m6")}
    return cfg, rc, sd, engines


def gen(eng, x, nb, ml, **kw):
    return eng.generate(x.cuda(), PROMPT, EOS, EOS, max_length=ml, num_beams=nb, suppress_tokens=SUP,
                        begin_suppress_tokens=BSUP, **kw)


def gp(nb, ml):
    return R.GenParams(prompt=PROMPT, eos_token_id=EOS, pad_token_id=EOS, max_length=ml, num_beams=nb,
                       suppress_tokens=SUP, begin_suppress_tokens=BSUP)


NEAR_TIE = 5e-5      # |score difference| (HF beam score: mean log-prob) under which two hypotheses count as tied


def assert_equally_scored(sd, rc, x_i, params, got_tokens, want_tokens, where):
    """A beam result that differs from the oracle's must be a hypothesis the ORACLE scores the same (to fp32 summation-order
    noise) as its own choice: seeded-random weights give nearly flat distributions, so beams that differ in one token can
    tie to ~1e-6 and either arithmetic may rank them either way.  Anything else is a real disagreement and fails."""
    s_got = R.score_sequence(sd, rc, x_i, params, got_tokens)
    s_want = R.score_sequence(sd, rc, x_i, params, [int(t) for t in want_tokens])
    assert abs(s_got - s_want) <= NEAR_TIE, (where, s_got, s_want, list(got_tokens), [int(t) for t in want_tokens])
    return abs(s_got - s_want)


def test_two_layer_8_windows_vs_oracle(gpu_lib, two_layer):
    cfg, rc, sd, engines = two_layer
    x = feats(8, seed=3)
    want_enc = R.encoder_forward(sd, rc, x)
    for dt, tol in (("f32", 5e-4), ("bf16", 8e-2), ("f16", 1.5e-2)):
        got = engines[dt].encode(x.cuda()).float().cpu()
        assert (got - want_enc).abs().max().item() <= tol * max(1.0, want_enc.abs().max().item()), dt
    for nb in (1, 4):
        want_tok, want_logits = R.generate(sd, rc, x, gp(nb, 10), return_first_logits=True)
        toks, lens, got = gen(engines["f32"], x, nb, 10, return_first_logits=True)
        assert (got.cpu() - want_logits).abs().max().item() <= 1e-3
        toks, lens = toks.cpu().numpy(), lens.cpu().numpy()
        same = [R.canonical(toks[i, :lens[i]].tolist(), 3, EOS, PROMPT) == R.canonical(want_tok[i].tolist(), 3, EOS, PROMPT) for i in range(8)]
        assert all(int(toks[i, 3]) == int(want_tok[i][3]) for i in range(8)), nb
        if nb == 1:            # f32 mode, greedy: token-exact
            assert all(same), same
        else:                  # beams: token-exact, or an equally scored hypothesis (asserted, not excused)
            for i in range(8):
                if not same[i]:
                    assert_equally_scored(sd, rc, x[i:i + 1], gp(nb, 10), toks[i, :lens[i]].tolist(), want_tok[i].tolist(), ("8w", nb, i))
            assert sum(same) >= 4, same
        for dt, cos_min, rel in (("bf16", 0.999, 0.1), ("f16", 0.99999, 0.015)):
            _, _, got16 = gen(engines[dt], x, nb, 10, return_first_logits=True)
            got16 = got16.cpu()
            assert torch.nn.functional.cosine_similarity(got16, want_logits, dim=1).min().item() > cos_min, dt
            assert (got16 - want_logits).abs().max().item() <= rel * max(1.0, want_logits.abs().max().item()), dt
            if nb == 4:
                assert torch.equal(got16[0::4], got16[1::4]) and torch.equal(got16[0::4], got16[3::4])
        # split-precision modes (8 slots x 20 heads: the few-slot cross-attention, 4 workgroups per (slot, head) over 24-bit K / V)
        for dt, rel in (("f16x3", 1e-4), ("bf16x3", 1e-3), ("f16m6", 1e-3)):
            t3, l3, got3 = gen(engines[dt], x, nb, 10, return_first_logits=True)
            assert (got3.cpu() - want_logits).abs().max().item() <= rel * max(1.0, want_logits.abs().max().item()), dt
            assert all(int(t3[i, 3]) == int(want_tok[i][3]) for i in range(8)), (dt, nb)


def test_two_layer_256_windows_1024_rows_vs_oracle(gpu_lib, two_layer):
    """The bench's row counts.  The oracle decodes 6 of the 256 windows (first / last / interior); the exact-parity f32 mode
    must reproduce its tokens and logits for them, the bf16 mode its logits within tolerance; every window of the bf16 run
    must agree with the f32 run at the logit level (cosine), which extends the oracle check to all 256 windows."""
    cfg, rc, sd, engines = two_layer
    x = feats(256, seed=5)
    pick = [0, 1, 77, 128, 200, 255]
    want_tok, want_logits = R.generate(sd, rc, x[pick], gp(4, 8), return_first_logits=True)
    t32, l32, g32 = gen(engines["f32"], x, 4, 8, return_first_logits=True)
    t16, l16, g16 = gen(engines["bf16"], x, 4, 8, return_first_logits=True)
    th, lh, gh = gen(engines["f16"], x, 4, 8, return_first_logits=True)
    g32, g16, gh = g32.cpu(), g16.cpu(), gh.cpu()
    assert torch.nn.functional.cosine_similarity(gh, g32, dim=1).min().item() > 0.99999
    assert float((th.cpu()[:, 3] == t32.cpu()[:, 3]).float().mean()) >= 0.97
    rows = [4 * p + j for p in pick for j in range(4)]
    assert (g32[rows] - want_logits).abs().max().item() <= 1e-3
    t32n, l32n = t32.cpu().numpy(), l32.cpu().numpy()
    same = [R.canonical(t32n[p, :l32n[p]].tolist(), 3, EOS, PROMPT) == R.canonical(want_tok[k].tolist(), 3, EOS, PROMPT)
            for k, p in enumerate(pick)]
    assert all(int(t32n[p, 3]) == int(want_tok[k][3]) for k, p in enumerate(pick))
    for k, p in enumerate(pick):         # token-exact, or an equally scored hypothesis (see the 8-window test)
        if not same[k]:
            assert_equally_scored(sd, rc, x[p:p + 1], gp(4, 8), t32n[p, :l32n[p]].tolist(), want_tok[k].tolist(), ("256w", p))
    assert sum(same) >= 3, same
    # greedy is free of beam ties: token-exact vs the oracle at 256 rows
    want_g = R.generate(sd, rc, x[pick], gp(1, 8))
    tg, lg = (v.cpu().numpy() for v in gen(engines["f32"], x, 1, 8))
    for k, p in enumerate(pick):
        assert R.canonical(tg[p, :lg[p]].tolist(), 3, EOS, PROMPT) == R.canonical(want_g[k].tolist(), 3, EOS, PROMPT), p
    assert torch.nn.functional.cosine_similarity(g16[rows], want_logits, dim=1).min().item() > 0.999
    assert (g16[rows] - want_logits).abs().max().item() <= 0.1 * max(1.0, want_logits.abs().max().item())
    # all 1024 rows: bf16 vs f32 mode
    assert torch.nn.functional.cosine_similarity(g16, g32, dim=1).min().item() > 0.999
    assert torch.equal(g16[0::4], g16[2::4])
    assert l16.cpu().tolist() == [8] * 256 or all(3 < v <= 8 for v in l16.cpu().tolist())
    # first generated token: bf16 may flip near-ties of these random-weight logits, but not often
    agree = float((t16.cpu()[:, 3] == t32.cpu()[:, 3]).float().mean())
    assert agree >= 0.9, agree
    # split-precision modes (f16x3 = the product default and the bench headline) at the same row counts: logits of ALL 1024 rows within
    # 1e-3 of the logit scale of the f32 mode (measured ~1e-5 / ~1e-4) and of the oracle on its subset, every first token equal
    scale = max(1.0, want_logits.abs().max().item())
    for dt, rel in (("f16x3", 1e-4), ("bf16x3", 1e-3), ("f16m6", 1e-3)):
        t3, l3, g3 = gen(engines[dt], x, 4, 8, return_first_logits=True)
        g3, t3 = g3.cpu(), t3.cpu()
        assert (g3 - g32).abs().max().item() <= rel * scale, dt
        assert (g3[rows] - want_logits).abs().max().item() <= 1e-3 * scale, dt
        assert torch.equal(g3[0::4], g3[2::4]), dt
        assert torch.equal(t3[:, 3], t32.cpu()[:, 3]), dt
        tg3, lg3 = (v.cpu().numpy() for v in gen(engines[dt], x, 1, 8))
        for k, p in enumerate(pick):      # greedy: token-exact vs the oracle
            assert R.canonical(tg3[p, :lg3[p]].tolist(), 3, EOS, PROMPT) == R.canonical(want_g[k].tolist(), 3, EOS, PROMPT), (dt, p)


def test_two_layer_1024_windows_4096_rows(gpu_lib, two_layer):
    """The engine's default slot count (1024 windows x 4 beams = 4096 decode rows): the decoder-step GEMMs then run on the
    large-tile kernels (q|k|v and fc1 on the 256x256 ping-pong kernel with their decode epilogues) instead of the split-K
    stream family of the 1024-row test.  First-step logits of all 4096 rows against the exact-parity f32 mode (run 256
    windows at a time: it does not depend on the slot count) and against the same 16-bit engine at 256 slots."""
    cfg, rc, sd, engines = two_layer
    x = feats(1024, seed=11)
    ref_logits, ref_tok = [], []
    for lo in range(0, 1024, 256):
        t, l, g = gen(engines["f32"], x[lo:lo + 256], 4, 8, return_first_logits=True, n_slots=256)
        ref_logits.append(g.cpu()); ref_tok.append(t.cpu())
    g32, t32 = torch.cat(ref_logits), torch.cat(ref_tok)
    for dt, cos_min in (("bf16", 0.999), ("f16", 0.99999)):
        t, l, g = gen(engines[dt], x, 4, 8, return_first_logits=True, n_slots=1024)
        assert engines[dt].last_stats()["n_slots"] == 1024
        g, t = g.cpu(), t.cpu()
        assert torch.nn.functional.cosine_similarity(g, g32, dim=1).min().item() > cos_min, dt
        assert torch.equal(g[0::4], g[2::4])                      # the beams of a window are identical at the first step
        assert float((t[:, 3] == t32[:, 3]).float().mean()) >= (0.9 if dt == "bf16" else 0.97), dt
        small = []
        for lo in range(0, 1024, 256):
            small.append(gen(engines[dt], x[lo:lo + 256], 4, 8, return_first_logits=True, n_slots=256)[2].cpu())
        small = torch.cat(small)
        assert torch.nn.functional.cosine_similarity(g, small, dim=1).min().item() > (0.9999 if dt == "bf16" else 0.999999), dt
        # run to run identical at this row count
        t2, l2 = gen(engines[dt], x, 4, 8, n_slots=1024)
        assert torch.equal(t2.cpu(), t) and torch.equal(l2.cpu(), l.cpu())
    # split-precision modes at 4096 rows: the x3 split-K ping-pong fc2, the 256x256 x3 decode epilogues and the 24-bit
    # cross-K/V kernel at 1024 slots — logits of all 4096 rows within 1e-3 of the scale of the f32 mode, first tokens equal,
    # and identical to the same engine at 256 slots wherever the GEMM plan family is row-count independent (reported otherwise)
    scale = max(1.0, g32.abs().max().item())
    for dt, rel in (("f16x3", 1e-4), ("bf16x3", 1e-3), ("f16m6", 1e-3)):
        t, l, g = gen(engines[dt], x, 4, 8, return_first_logits=True, n_slots=1024)
        assert engines[dt].last_stats()["n_slots"] == 1024
        g, t = g.cpu(), t.cpu()
        assert (g - g32).abs().max().item() <= rel * scale, dt
        assert torch.equal(g[0::4], g[2::4]), dt
        assert torch.equal(t[:, 3], t32[:, 3]), dt
        t2, l2 = gen(engines[dt], x, 4, 8, n_slots=1024)
        assert torch.equal(t2.cpu(), t) and torch.equal(l2.cpu(), l.cpu())
        small_t = torch.cat([gen(engines[dt], x[lo:lo + 256], 4, 8, n_slots=256)[0].cpu() for lo in range(0, 1024, 256)])
        assert torch.equal(small_t, t), dt        # the same tokens at 256 and at 1024 slots
        # ... and at 512 slots (2 048 rows: r05 plans — fc1 on one 62 %-full round of 256x256 tiles, q|k|v as two split-K copies)
        mid_t = torch.cat([gen(engines[dt], x[lo:lo + 512], 4, 8, n_slots=512)[0].cpu() for lo in range(0, 1024, 512)])
        assert torch.equal(mid_t, t), dt


def test_two_layer_tokens_do_not_depend_on_the_admission(gpu_lib, two_layer):
    """in the split modes the admission's pass (forced prompt + first generated step) and the cross-K/V GEMMs used to
    run on GEMM plans that followed the ADMISSION size — which depends on when a window's neighbours finished.  Since r06 they run on
    plans fixed by the call (GemmArgs::plan_m: the decode step's plan for the pass, a full encoder chunk's for the cross K / V) and
    the top-k slices the vocabulary as a decode step does.  Flat random-weight logits are the sensitive probe (top-1 / top-2 margins
    down to summation-order noise): 40 windows with uneven length caps through 16 slots — the same windows in reversed order and
    with single-slot refills (admissions of 1-3 windows instead of 2+) meet other neighbours in admissions of other sizes and must
    decode to the same tokens."""
    cfg, rc, sd, engines = two_layer
    x = feats(40, seed=17)
    caps = (torch.randint(4, 12, (40,), generator=torch.Generator().manual_seed(9)) + 3).to(torch.int32)
    perm = torch.arange(39, -1, -1)
    for dt in ("f16m6", "f16x3", "bf16x3", "f32"):
        eng = engines[dt]
        a_t, a_l = (v.cpu() for v in gen(eng, x, 4, 14, n_slots=16, window_max_length=caps))
        assert eng.last_stats()["n_admissions"] >= 3, dt
        b_t, b_l = (v.cpu() for v in gen(eng, x[perm], 4, 14, n_slots=16, window_max_length=caps[perm]))
        assert torch.equal(b_l, a_l[perm]) and torch.equal(b_t, a_t[perm]), dt
        c_t, c_l = (v.cpu() for v in gen(eng, x, 4, 14, n_slots=16, window_max_length=caps, refill_min=1))
        assert eng.last_stats()["n_admissions"] > 6, dt
        assert torch.equal(c_l, a_l) and torch.equal(c_t, a_t), dt


@pytest.fixture(scope="module")
def full_large():
    from whisperseg_amd.engine import Engine
    return Engine.random(large_cfg(32), "cuda:0", "bf16", seed=0)


@pytest.mark.parametrize("dt", ["bf16", "f16m6"])
@pytest.mark.parametrize("n", [8, 120])
def test_full_large_properties(gpu_lib, full_large, n, dt):
    """configs[2] (large x 8) and the single-GPU share of configs[3] (120 windows) at full depth, in plain bf16 and in the product
    default f16m6 (same bf16-representable weights)."""
    from whisperseg_amd.engine import Engine
    eng = full_large if dt == "bf16" else full_large.sibling(dt)
    x = feats(n, seed=7 + n)
    a = gen(eng, x, 4, 3 + 12, return_first_logits=True)
    b = gen(eng, x, 4, 3 + 12, return_first_logits=True)
    for u, v in zip(a, b):                      # determinism
        assert torch.equal(u, v)
    toks, lens, logits = (t.cpu() for t in a)
    assert torch.isfinite(logits).all()
    assert torch.equal(logits[0::4], logits[1::4]) and torch.equal(logits[0::4], logits[3::4])    # beams equal at step 1
    assert ((toks[:, :3] == torch.tensor(PROMPT)).all())
    sup = torch.tensor(SUP)
    assert not torch.isin(toks[:, 3:], sup).any()                   # suppressed ids never generated
    assert not torch.isin(toks[:, 3], torch.tensor(BSUP)).any()     # begin-suppressed ids never first
    perm = torch.randperm(n, generator=torch.Generator().manual_seed(1))
    tp, lp, gp_ = (t.cpu() for t in gen(eng, x[perm], 4, 3 + 12, return_first_logits=True))
    assert torch.equal(gp_[0::4], logits[0::4][perm])               # a window's result does not depend on its slot
    assert torch.equal(tp, toks[perm]) and torch.equal(lp, lens[perm])


@pytest.fixture(scope="module")
def full_large_f32():
    """32 + 32 layers with FULL-MANTISSA fp32 weights in HF naming (so that the CPU oracle can score hypotheses on exactly the values the
    engines compute with), the exact-parity f32 engine over them and its decode of 16 windows."""
    from whisperseg_amd.engine import Engine
    cfg = large_cfg(32)
    rc = R.RefConfig.from_hf_dict(cfg)
    sd = R.random_state_dict(rc, seed=5, fast=True)
    f32 = Engine.from_state_dict(sd, cfg, "cuda:0", "f32")
    x = feats(16, seed=33)
    t32, l32, g32 = (t.cpu() for t in gen(f32, x, 4, 12, return_first_logits=True))
    return cfg, rc, sd, f32, x, t32, l32, g32


@pytest.mark.parametrize("dt,rel", [("f16x3", 1e-4), ("bf16x3", 5e-4), ("f16m6", 5e-4)])
def test_full_large_split_precision_vs_f32_mode(gpu_lib, full_large_f32, dt, rel):
    """32 + 32 layers, 16 windows: the split-precision modes (product default f16x3; bf16x3; f16m6) against the exact-parity f32 mode on the
    same fp32 weights.  Encoder output and first-step logits within rel x scale (measured 2.8e-5 / 1.1e-4 / 3.0e-4 of the logit scale).
    WHOLE beam-search results (12 positions, 4 beams): identical to the f32 mode's, or — these are random weights, their distributions are
    flat and hypotheses whose total scores tie to ~1e-4 exist — a hypothesis the CPU ORACLE scores the same as the f32 mode's choice to
    within the mode's own measured logit error (r06, VERDICT r05 item 1b: the differing windows are ASSERTED near-ties through
    oracle.whisper_ref.score_sequence on the fp32 weights, as assert_equally_scored does at 2 layers; anything else fails).  At most a
    quarter of the windows may differ at all."""
    cfg, rc, sd, f32, x, t32, l32, g32 = full_large_f32
    x3 = f32.sibling(dt)
    e3, e32 = x3.encode(x[:4].cuda()).float(), f32.encode(x[:4].cuda())
    assert (e3 - e32).abs().max().item() <= rel * max(1.0, e32.abs().max().item())
    t3, l3, g3 = (t.cpu() for t in gen(x3, x, 4, 12, return_first_logits=True))
    err = (g3 - g32).abs().max().item()
    bound = rel * max(1.0, g32.abs().max().item())
    assert err <= bound, (err, bound)
    # greedy choice of every beam row at the first step: equal, or a tie in the f32 mode's OWN logits to within twice the asserted error
    a3, a32 = g3.argmax(dim=1), g32.argmax(dim=1)
    for r_ in (a3 != a32).nonzero().flatten().tolist():
        assert abs(float(g32[r_, a3[r_]] - g32[r_, a32[r_]])) <= 2 * bound, (r_, int(a3[r_]), int(a32[r_]))
    n = t3.shape[0]
    differing = [w_ for w_ in range(n) if not (torch.equal(t3[w_], t32[w_]) and int(l3[w_]) == int(l32[w_]))]
    print(dt, "first-step logit error", err, "windows whose beam result differs from the f32 mode:", differing)
    assert len(differing) <= n // 4, differing
    # A hypothesis score is a mean of log-probabilities, each of which moves by at most ~2 x the logit error (the logit and the
    # log-sum-exp): two hypotheses may swap when the oracle's scores are within 4 x the mode's measured error.
    tie = 4 * max(err, 2e-5)
    for w_ in differing:
        s_got = R.score_sequence(sd, rc, x[w_:w_ + 1], gp(4, 12), t3[w_, :int(l3[w_])].tolist())
        s_want = R.score_sequence(sd, rc, x[w_:w_ + 1], gp(4, 12), t32[w_, :int(l32[w_])].tolist())
        assert abs(s_got - s_want) <= tie, (dt, w_, s_got, s_want, tie)


def test_full_large_bf16_vs_f32_mode(gpu_lib, full_large):
    """32 + 32 layers: the production bf16 path against the exact-parity f32 mode of the same kernels on the same
    (bf16-representable) weights — the f32 mode itself is pinned on the oracle by the 2-layer tests above and on HF by the
    golden tests."""
    from whisperseg_amd.engine import Engine
    eng = full_large
    f32 = Engine(eng.geo, {k: v.float() for k, v in eng.weights.items()}, eng.device, "f32")
    x = feats(4, seed=21)
    e16 = eng.encode(x.cuda()).float()
    e32 = f32.encode(x.cuda())
    assert (e16 - e32).abs().max().item() <= 0.15 * max(1.0, e32.abs().max().item())
    assert torch.nn.functional.cosine_similarity(e16.flatten(1), e32.flatten(1), dim=1).min().item() > 0.995
    _, _, g16 = gen(eng, x, 4, 8, return_first_logits=True)
    _, _, g32 = gen(f32, x, 4, 8, return_first_logits=True)
    assert torch.nn.functional.cosine_similarity(g16, g32, dim=1).min().item() > 0.995
    assert (g16 - g32).abs().max().item() <= 0.15 * max(1.0, g32.abs().max().item())

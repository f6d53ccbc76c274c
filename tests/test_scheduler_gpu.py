"""In-flight batching (wseg_generate's slot scheduler) on the GPU.

The reference decodes batch by batch and every batch runs until its slowest window has finished (model.py:653); folder
mode is a serial loop (scripts/segment.py:39-56).  The engine decodes any number of windows through a fixed number of
window slots, refilling a slot as soon as its window ends.  Properties checked here (windows are independent):
  * tokens of a window do not depend on the slot count, the refill order or what ran beside it — exact in f32 mode, and
    bit-identical in bf16 mode whenever the slot count (= GEMM row count) is the same;
  * a trained model that emits EOS after 10-40 tokens with max_length = 448 executes about as many decode steps as its
    longest window needs, not 445 (device-side idle flags + bounded host look-ahead);
  * per-window length caps behave exactly like separate calls with that max_length;
  * the scheduler statistics add up (every window decoded once, occupancy within (0, 1]);
  * a caller stream other than the default one, a sampling call and a per-window-cap call reuse the captured step graph."""
import json
import os

import numpy as np
import pytest
import torch

import golden_inputs as GI
from conftest import GOLDEN
from tools import tiny_model as TM

pytestmark = pytest.mark.gpu
MODEL_DIR = os.path.join(GOLDEN, "tiny_model")


def tiny_engine(dtype):
    from safetensors.torch import load_file
    from whisperseg_amd.engine import Engine
    sd = {k: v.float() for k, v in load_file(os.path.join(MODEL_DIR, "model.safetensors")).items()}
    with open(os.path.join(MODEL_DIR, "config.json")) as f:
        cfg = json.load(f)
    return Engine.from_state_dict(sd, cfg, "cuda:0", dtype)


def tiny_feats(n_recordings, seed0=300):
    from whisperseg_amd.audio_utils import WhisperSegFeatureExtractor
    ext = WhisperSegFeatureExtractor(TM.SR, TM.STS, device="cuda:0")
    out = []
    for s in range(n_recordings):
        clip = TM.synth_clip(np.random.default_rng(seed0 + s))[0]
        out.append(ext.extract_windows(torch.from_numpy(clip).cuda(), torch.tensor([0]), len(clip))[0])
    return torch.stack(out)


def gen(eng, x, nb=4, ml=448, **kw):
    t, l = eng.generate(x, TM.PROMPT, TM.EOT, TM.EOT, max_length=ml, num_beams=nb, suppress_tokens=TM.SUPPRESS,
                        begin_suppress_tokens=TM.BEGIN_SUPPRESS, **kw)
    return t.cpu(), l.cpu()


@pytest.mark.parametrize("dtype", ["f32", "f16x3", "bf16x3", "f16m6", "bf16", "f16"])
@pytest.mark.parametrize("nb", [1, 4])
def test_slot_refill_gives_the_same_tokens(gpu_lib, dtype, nb):
    eng = tiny_engine(dtype)
    x = tiny_feats(23)
    ref_t, ref_l = gen(eng, x, nb)                       # one slot per window, all start together
    assert len(set(ref_l.tolist())) > 3                  # EOS fires at varied lengths: the refill order is non-trivial
    for slots, refill in ((5, 1), (5, 0), (8, 3), (1, 0)):
        t, l = gen(eng, x, nb, n_slots=slots, refill_min=refill)
        st = eng.last_stats()
        assert st["n_windows"] == 23 and st["n_slots"] == slots and st["n_admissions"] >= -(-23 // slots)
        assert 0 < st["occupancy"] <= 1.0
        if dtype in ("f32", "f16x3", "bf16x3", "f16m6"):
            # f32: every dot product is one k-ordered chain whatever the plan -> bit-identical.  Split-precision modes (f16x3 is
            # the product default): split-K plans follow the row count, which moves logits by fp32 summation-order noise (~1e-7 of
            # their scale) — four orders of magnitude below the smallest top-1 / top-2 margin of the parity sweep (1e-5,
            # profiles/history/r03_precision_study.json "margins") — so the TOKENS must not depend on the slot count 1 / 5 / 8 / 23
            assert torch.equal(l, ref_l) and torch.equal(t, ref_t), (dtype, slots, refill)
        if dtype != "f32":   # 16-bit GEMM operands: results are bit-stable for a FIXED slot count whatever ran beside a window
            t2, l2 = gen(eng, x.flip(0), nb, n_slots=slots, refill_min=refill)
            assert torch.equal(l2.flip(0), l) and torch.equal(t2.flip(0), t), (dtype, slots, refill)
    # same slot count, different neighbours: decode the windows 6 at a time in 6 slots vs all 23 through 6 slots
    t6, l6 = gen(eng, x, nb, n_slots=6)
    for lo in range(0, 18, 6):
        tb, lb = gen(eng, x[lo:lo + 6], nb, n_slots=6)
        assert torch.equal(tb, t6[lo:lo + 6]) and torch.equal(lb, l6[lo:lo + 6]), (dtype, lo)


@pytest.mark.parametrize("dtype", ["f32", "f16x3", "f16m6", "bf16"])
def test_non_default_stream_and_graph_reuse(gpu_lib, dtype):
    """The call is stream-ordered on the caller's stream; the captured decode-step graph survives calls that differ only in
    per-call data (sampling seed, per-window length caps: both live in device memory / the admission kernel)."""
    eng = tiny_engine(dtype)
    x = tiny_feats(23)
    ref_t, ref_l = gen(eng, x, 4, n_slots=5)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        t, l = eng.generate(x.clone(), TM.PROMPT, TM.EOT, TM.EOT, max_length=448, num_beams=4, suppress_tokens=TM.SUPPRESS,
                            begin_suppress_tokens=TM.BEGIN_SUPPRESS, n_slots=5)
        t, l = t.cpu(), l.cpu()
    assert torch.equal(l, ref_l) and torch.equal(t, ref_t)
    caps = [6 + (i % 5) for i in range(23)]
    tc, lc = gen(eng, x, 4, n_slots=5, window_max_length=caps)
    assert all(int(v) <= c for v, c in zip(lc.tolist(), caps))
    t2, l2 = gen(eng, x, 4, n_slots=5)                   # back to uncapped: same graph, same tokens as before
    assert torch.equal(l2, ref_l) and torch.equal(t2, ref_t)
    # sampling with two seeds: reproducible per seed, different across seeds, and the beam call after it is unaffected
    a1 = gen(eng, x, 1, n_slots=5, top_k=8, top_p=0.95, seed=1)
    a2 = gen(eng, x, 1, n_slots=5, top_k=8, top_p=0.95, seed=1)
    b1 = gen(eng, x, 1, n_slots=5, top_k=8, top_p=0.95, seed=2)
    assert torch.equal(a1[0], a2[0]) and torch.equal(a1[1], a2[1])
    assert not torch.equal(a1[0], b1[0])
    t3, l3 = gen(eng, x, 4, n_slots=5)
    assert torch.equal(l3, ref_l) and torch.equal(t3, ref_t)


@pytest.mark.parametrize("dtype", ["f32", "f16x3", "bf16x3", "f16m6", "bf16", "f16"])
def test_default_slot_count_with_a_long_queue(gpu_lib, dtype):
    """1 100 windows (23 distinct recordings repeated) through the engine's default 1 024 slots: copies of a window give the
    same tokens wherever they ran (first batch, refilled slot, drain), in every mode; in f32 mode they also equal the
    one-slot-per-window result of the 23 originals."""
    from whisperseg_amd.engine import DEFAULT_SLOTS
    eng = tiny_engine(dtype)
    base = tiny_feats(23)
    x = base.repeat(48, 1, 1)[:1100]
    t, l = gen(eng, x, 4, n_slots=None)
    st = eng.last_stats()
    assert st["n_slots"] == min(DEFAULT_SLOTS, 1100) and st["n_windows"] == 1100
    for i in range(23, 1100):
        assert int(l[i]) == int(l[i % 23]) and torch.equal(t[i], t[i % 23]), (dtype, i)
    if dtype in ("f32", "f16x3", "bf16x3", "f16m6"):       # ... and in the split-precision modes (23 slots against 1 024: see the refill test)
        ref_t, ref_l = gen(eng, base, 4)
        assert torch.equal(l[:23], ref_l) and torch.equal(t[:23], ref_t)


@pytest.mark.parametrize("dtype", ["bf16", "f16x3", "f16m6"])
def test_early_stop_executes_few_steps(gpu_lib, dtype):
    """with max_length 448 and EOS after 10-40 tokens the GPU used to run every queued step at full cost."""
    eng = tiny_engine(dtype)
    x = tiny_feats(12, seed0=500)
    t, l = gen(eng, x, 4, 448)
    longest = int(l.max())
    assert longest < 120
    steps = eng.last_stats()["n_steps"]
    # a window of total length n needs n - 1 fed positions (+ the beam heuristic may run a few steps past the first EOS);
    # the host may run `lookahead` = 2 steps ahead
    assert steps <= longest + 8, (steps, longest)
    assert eng.last_timing()[3] == steps


@pytest.mark.parametrize("dtype", ["f32", "f16x3", "f16m6"])
def test_per_window_length_caps(gpu_lib, dtype):
    eng = tiny_engine(dtype)
    x = tiny_feats(9, seed0=700)
    caps = [6, 448, 12, 9, 448, 5, 20, 7, 448]
    t, l = gen(eng, x, 4, 448, window_max_length=caps, n_slots=4)
    for i, cap in enumerate(caps):
        ti, li = gen(eng, x[i:i + 1], 4, cap)
        n = int(li[0])
        assert int(l[i]) == n and torch.equal(t[i, :n], ti[0, :n]), i
        assert n <= cap


def test_more_windows_than_slots_bounds_the_workspace(gpu_lib):
    """The workspace is sized by the slot count, not by the number of windows."""
    eng = tiny_engine("bf16")
    x = tiny_feats(4, seed0=900).repeat(16, 1, 1)        # 64 windows
    gen(eng, x, 4, 448, n_slots=8)
    ws8 = eng._ws.numel()
    assert ws8 == eng.lib.wseg_workspace_bytes(eng.handle, 8, 4, 448) or ws8 >= eng.lib.wseg_workspace_bytes(eng.handle, 8, 4, 448)
    assert eng.lib.wseg_workspace_bytes(eng.handle, 64, 4, 448) > 4 * eng.lib.wseg_workspace_bytes(eng.handle, 8, 4, 448)
    t, l = gen(eng, x, 4, 448, n_slots=8)
    for r in range(1, 16):                               # identical windows decode identically wherever they ran
        assert torch.equal(t[4 * r:4 * r + 4], t[:4]) and torch.equal(l[4 * r:4 * r + 4], l[:4])


def test_sampling_top_k_top_p(gpu_lib):
    """num_beams == 1 with top_k > 1 (reference model.py:615-616: do_sample): reproducible for a seed, seed-dependent, and
    the first sampled token follows softmax over the top_k processed logits cut to the nucleus top_p (HF TopK + TopP
    warpers) — checked against the first-step logits over 600 seeds."""
    eng = tiny_engine("f32")
    x = tiny_feats(3, seed0=40)
    kw = dict(max_length=6, num_beams=1, suppress_tokens=TM.SUPPRESS, begin_suppress_tokens=TM.BEGIN_SUPPRESS)
    _, _, logits = eng.generate(x, TM.PROMPT, TM.EOT, TM.EOT, return_first_logits=True, **kw)
    logits = logits.cpu().double()
    logits[:, TM.SUPPRESS] = -float("inf")
    logits[:, TM.BEGIN_SUPPRESS] = -float("inf")
    a = eng.generate(x, TM.PROMPT, TM.EOT, TM.EOT, top_k=5, top_p=0.9, seed=7, **kw)[0].cpu()
    b = eng.generate(x, TM.PROMPT, TM.EOT, TM.EOT, top_k=5, top_p=0.9, seed=7, **kw)[0].cpu()
    assert torch.equal(a, b)
    for top_k, top_p in ((4, 1.0), (6, 0.8)):
        vals, idx = logits.topk(top_k, dim=1)
        probs = torch.softmax(vals, dim=1)
        keep = (probs.cumsum(1) - probs) < top_p                 # mass before the candidate < top_p
        keep[:, 0] = True
        want = torch.where(keep, probs, torch.zeros_like(probs))
        want = want / want.sum(1, keepdim=True)
        counts = torch.zeros_like(want)
        n_draws = 600
        for seed in range(n_draws):
            t = eng.generate(x, TM.PROMPT, TM.EOT, TM.EOT, top_k=top_k, top_p=top_p, seed=seed, **kw)[0].cpu()
            for i in range(3):
                hit = (idx[i] == int(t[i, 3])).nonzero()
                assert len(hit) == 1, (top_k, top_p, i, int(t[i, 3]))          # only top_k candidates are ever drawn
                counts[i, hit[0, 0]] += 1
        freq = counts / n_draws
        assert (freq[~keep] == 0).all()                          # outside the nucleus: never
        assert (freq - want).abs().max().item() < 0.07, (freq, want)   # 600 draws: 3.4 sigma of a p = 0.5 proportion
    with pytest.raises(Exception):
        eng.generate(x, TM.PROMPT, TM.EOT, TM.EOT, top_k=17, **kw)


def test_segment_with_sampling_is_seeded_by_torch(gpu_lib):
    from whisperseg_amd.model import WhisperSegmenter
    seg = WhisperSegmenter(MODEL_DIR, device="cuda", device_ids=[0], dtype="f32")
    audio = GI.tiny_recording(100, 3)
    torch.manual_seed(5)
    a = seg.segment(audio, TM.SR, num_beams=1, top_k=3, top_p=0.95)
    torch.manual_seed(5)
    b = seg.segment(audio, TM.SR, num_beams=1, top_k=3, top_p=0.95)
    assert a == b and set(a) == {"onset", "offset", "cluster"}
    with pytest.raises(NotImplementedError):
        seg.segment(audio, TM.SR, num_beams=1, top_k=50)


def test_lookahead_and_eager_steps_give_the_same_tokens(gpu_lib):
    """The host may run 1..6 steps ahead of the device (statuses are consumed behind events), and WSEG_NO_GRAPH=1 launches
    every step eagerly instead of replaying the captured graph: neither may change a token."""
    import subprocess
    import sys
    import tempfile
    from conftest import ROOT
    eng = tiny_engine("f32")
    x = tiny_feats(11, seed0=1200)
    ref_t, ref_l = gen(eng, x, 4, n_slots=4)
    for la in (1, 3, 6, 50):
        t, l = gen(eng, x, 4, n_slots=4, lookahead=la)
        assert torch.equal(t, ref_t) and torch.equal(l, ref_l), la
    code = r'''
import sys, torch
sys.path.insert(0, sys.argv[2]); sys.path.insert(0, sys.argv[2] + "/tests")
import test_scheduler_gpu as T
eng = T.tiny_engine("f32")
x = T.tiny_feats(11, seed0=1200)
torch.save(T.gen(eng, x, 4, n_slots=4), sys.argv[1])
'''
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "eager.pt")
        subprocess.check_call([sys.executable, "-c", code, path, ROOT], env={**os.environ, "WSEG_NO_GRAPH": "1"})
        t, l = torch.load(path)
    assert torch.equal(t, ref_t) and torch.equal(l, ref_l)


def test_slot_cap_and_workspace_lifecycle(gpu_lib):
    """`batch_size` no longer bounds the decode memory: `segmenter.max_slots` / n_slots does, and the engine gives a
    much larger workspace back when a later call needs less than a quarter of it."""
    import os
    from conftest import GOLDEN
    from whisperseg_amd.model import WhisperSegmenter
    eng = tiny_engine("f32")
    x = tiny_feats(23)
    ref_t, ref_l = gen(eng, x, 4)
    big = eng._ws.numel()
    assert eng.last_stats()["n_slots"] == 23
    t, l = gen(eng, x, 4, n_slots=2)                      # 2 slots need less than a quarter of 23 slots' workspace ...
    assert eng.last_stats()["n_slots"] == 2
    assert torch.equal(l, ref_l) and torch.equal(t, ref_t)
    assert eng._ws.numel() == big                         # ... but one small call never shrinks it (hysteresis)
    if big > (2 << 30):                                   # and only workspaces above 2 GiB are worth re-allocating at all
        for _ in range(4):
            gen(eng, x, 4, n_slots=2)
        assert eng._ws.numel() < big
    eng.release_workspace()
    assert eng._ws is None
    t, l = gen(eng, x, 4, n_slots=5)
    assert torch.equal(l, ref_l) and torch.equal(t, ref_t)
    seg = WhisperSegmenter(os.path.join(GOLDEN, "tiny_model"), device="cuda", device_ids=[0], dtype="f32")
    clip = TM.synth_clip(np.random.default_rng(4))[0]
    audio = np.concatenate([clip] * 4)
    want = seg.segment(audio, TM.SR)
    seg.max_slots = 1
    assert seg.segment(audio, TM.SR, batch_size=64) == want
    assert seg.model_list[0].last_stats()["n_slots"] == 1


# ---- paged self-attention K / V (ABI 5) ---------------------------------------------------------------------------------------
def _units(eng, slots, nb, ml, per):
    """workspace bytes for `per` pooled positions per slot on average"""
    return eng.lib.wseg_workspace_bytes_kv(eng.handle, slots, nb, ml, per)


@pytest.mark.parametrize("dtype", ["f32", "f16x3", "f16m6"])
def test_paged_kv_default_pool_equals_full_pool(gpu_lib, dtype):
    """max_length = 448 with the DEFAULT pool (64 positions per slot on average instead of 448 reserved per slot) gives exactly
    the tokens of a fully provisioned pool — paging only changes where a K / V row lives — in a fraction of the workspace."""
    eng = tiny_engine(dtype)
    x = tiny_feats(23)
    full_t, full_l = gen(eng, x, 4, 448, kv_positions=448)
    st = eng.last_stats()
    assert st["kv_units_total"] == 23 * 56 and st["n_preemptions"] == 0 and 0 < st["kv_units_peak"] <= st["kv_units_total"]
    peak_positions = st["kv_units_peak"] * 8
    assert peak_positions < 23 * 130                       # the trained model stops after 10-40 tokens: nowhere near 448 per slot
    eng.release_workspace()                                # (the pool is whatever the workspace has room for: start from a fresh one)
    t, l = gen(eng, x, 4, 448)                             # default pool
    st = eng.last_stats()
    assert st["kv_units_total"] == 23 * 8 and st["n_preemptions"] == 0
    assert torch.equal(t, full_t) and torch.equal(l, full_l)
    assert _units(eng, 23, 4, 448, 0) < _units(eng, 23, 4, 448, 448)
    assert _units(eng, 23, 4, 448, 0) == _units(eng, 23, 4, 448, 64) >= _units(eng, 23, 4, 64, 0)


@pytest.mark.parametrize("dtype", ["f32", "f16x3", "f16m6"])
@pytest.mark.parametrize("nb", [1, 4])
def test_paged_kv_short_pool_preempts_and_still_gives_the_same_tokens(gpu_lib, nb, dtype):
    """A pool far too small for the windows in flight: the scheduler preempts the youngest slot (device-side abort, window
    re-queued, decoded again from scratch later) instead of failing, the oldest window always progresses, and every window
    still gets exactly the tokens of the uncontended run (f32 mode: bit-exact whatever the slot history; the fast mode
    f16m6: abort + re-admission run through the M6-row operand writers, tokens asserted equal as in the refill test)."""
    eng = tiny_engine(dtype)
    x = tiny_feats(23)
    ref_t, ref_l = gen(eng, x, nb, 448, kv_positions=448)
    longest = int(ref_l.max())
    assert longest > 16
    # 8 slots sharing 8 positions per slot on average: fewer units (8) than ONE long window needs (ceil(longest / 8)) would be
    # refused, so give the pool exactly one slot's worth of max_length = 64 -> 8 units for 8 slots
    eng.release_workspace()
    t, l = gen(eng, x, nb, 64, n_slots=8, kv_positions=8)
    st = eng.last_stats()
    assert st["kv_units_total"] == 8 and st["n_preemptions"] > 0, st      # 8 slots x 1 page == one slot's worth of 64 positions
    assert st["kv_units_peak"] <= 8
    ref64_t, ref64_l = gen(eng, x, nb, 64, n_slots=8, kv_positions=64)
    assert eng.last_stats()["n_preemptions"] == 0
    assert torch.equal(t, ref64_t) and torch.equal(l, ref64_l)
    if longest <= 64:
        assert torch.equal(l, ref_l) and torch.equal(t[:, :64], ref_t[:, :64])
    # a milder shortage with the refill running: 23 windows through 5 slots with 16 positions per slot on average
    eng.release_workspace()
    t2, l2 = gen(eng, x, nb, 448, n_slots=5, kv_positions=16, refill_min=1)
    assert torch.equal(l2, ref_l) and torch.equal(t2, ref_t)
    assert eng.last_stats()["kv_units_total"] == 56                      # max(5 slots x 2 pages, one slot's worth of 448 positions)
    eng.release_workspace()
    t3, l3 = gen(eng, x, nb, 100, n_slots=8, kv_positions=16, refill_min=1)
    st3 = eng.last_stats()
    assert st3["kv_units_total"] == 16 and st3["kv_units_peak"] <= 16, st3      # admission control and / or preemption keep it inside the pool
    r3t, r3l = gen(eng, x, nb, 100, n_slots=8, kv_positions=100, refill_min=1)
    assert eng.last_stats()["n_preemptions"] == 0
    assert torch.equal(l3, r3l) and torch.equal(t3, r3t)


def test_paged_kv_workspace_too_small_is_an_error(gpu_lib):
    """Less than one slot's worth of pages (a lone window could not reach max_length) is refused, not deadlocked."""
    import ctypes as C
    from whisperseg_amd import _lib
    eng = tiny_engine("f32")
    x = tiny_feats(2)
    need = eng.lib.wseg_workspace_bytes(eng.handle, 2, 4, 448)
    ws = torch.empty(need // 4, dtype=torch.uint8, device="cuda:0")
    gp = _lib.GenerateParams()
    for i, tok in enumerate(TM.PROMPT):
        gp.prompt[i] = tok
    gp.prompt_len, gp.eos_token_id, gp.pad_token_id, gp.max_length, gp.num_beams, gp.length_penalty = 3, TM.EOT, TM.EOT, 448, 4, 1.0
    toks = torch.empty((2, 448), dtype=torch.int32, device="cuda:0")
    lens = torch.empty((2,), dtype=torch.int32, device="cuda:0")
    rc = eng.lib.wseg_generate(eng.handle, x.data_ptr(), 2, C.byref(gp), ws.data_ptr(), ws.numel(), toks.data_ptr(), lens.data_ptr(), None)
    assert rc == -3 and b"workspace too small" in eng.lib.wseg_last_error()


@pytest.mark.parametrize("dtype", ["f16m6", "f16x3", "bf16x3"])
@pytest.mark.parametrize("nb", [1, 2, 4])
def test_prompt_pass_equals_stepping_through_the_prompt(gpu_lib, dtype, nb, monkeypatch):
    """Split-precision modes run the forced prompt positions 0 .. P - 2 of every admission as ONE pass over the admitted windows
    (rows = windows x positions; the windows' cross-attention K / V are streamed once for them) and start the slots at position
    P - 1.  Same tokens as stepping through the prompt (test knob WSEG_NO_PROMPT_PASS), P - 1 fewer steps when all windows start
    together, and the same under refills (admissions of a few windows, every one with its own prompt pass)."""
    eng = tiny_engine(dtype)
    x = tiny_feats(23)
    res = {}
    for mode in ("pass", "step"):
        if mode == "step":
            monkeypatch.setenv("WSEG_NO_PROMPT_PASS", "1")
        for slots, refill in ((23, 0), (5, 1), (1, 0)):
            t, l = gen(eng, x, nb, n_slots=slots, refill_min=refill)
            res[mode, slots] = (t, l, eng.last_stats()["n_steps"])
    for slots in (23, 5, 1):
        assert torch.equal(res["pass", slots][0], res["step", 23][0]) and torch.equal(res["pass", slots][1], res["step", 23][1]), slots
        assert torch.equal(res["step", slots][0], res["step", 23][0])
    # the pass also runs the first generated step (one row per window: the beams are still copies): P fewer steps
    assert res["pass", 23][2] == res["step", 23][2] - len(TM.PROMPT)
    assert res["pass", 1][2] <= res["step", 1][2] - 23 * len(TM.PROMPT) + 8
    # ... and without that merge (test knob): P - 1 fewer, same tokens
    monkeypatch.delenv("WSEG_NO_PROMPT_PASS")
    monkeypatch.setenv("WSEG_NO_FIRST_STEP_MERGE", "1")
    for slots, refill in ((23, 0), (5, 1)):
        t, l = gen(eng, x, nb, n_slots=slots, refill_min=refill)
        assert torch.equal(t, res["step", 23][0]) and torch.equal(l, res["step", 23][1]), slots
        if slots == 23:
            assert eng.last_stats()["n_steps"] == res["step", 23][2] - (len(TM.PROMPT) - 1)


@pytest.mark.parametrize("mode", ["f16x3", "f16m6"])
@pytest.mark.parametrize("plen", [1, 2, 6, 8])
def test_prompt_pass_with_other_prompt_lengths(gpu_lib, plen, mode, monkeypatch):
    """The C-ABI takes prompts of 1..8 tokens: P <= 4: the pass covers the whole prompt and the first generated step; longer prompts: the
    first 4 positions, the rest is stepped through; same tokens as stepping through all of it."""
    eng = tiny_engine(mode)
    x = tiny_feats(7)
    prompt = (TM.PROMPT + [TM.PROMPT[1], TM.PROMPT[2]] * 3)[:plen]

    def run():
        t, l = eng.generate(x, prompt, TM.EOT, TM.EOT, max_length=40, num_beams=4, suppress_tokens=TM.SUPPRESS,
                            begin_suppress_tokens=TM.BEGIN_SUPPRESS, n_slots=3, refill_min=1)
        return t.cpu(), l.cpu(), eng.last_stats()["n_steps"]

    t1, l1, s1 = run()
    monkeypatch.setenv("WSEG_NO_PROMPT_PASS", "1")
    t2, l2, s2 = run()
    assert torch.equal(t1, t2) and torch.equal(l1, l2)
    assert s1 < s2 if plen <= 4 else s1 <= s2      # P <= 4: the pass also runs the first generated step
    assert (t1[:, :plen] == torch.tensor(prompt)).all()


@pytest.mark.parametrize("mode", ["f16x3", "f16m6"])
@pytest.mark.parametrize("nb", [1, 4])
def test_first_step_in_the_pass_with_windows_that_end_there(gpu_lib, nb, mode, monkeypatch):
    """Per-window caps of P + 1 .. P + 3: some windows are finished by the first generated step, which the admission's pass runs itself
    (one row per window) — they must retire with exactly that one token, refills included; same tokens as stepping through everything,
    also when the call's max_length leaves no step after the first (the merge is then off by construction)."""
    eng = tiny_engine(mode)
    x = tiny_feats(17)
    P = len(TM.PROMPT)
    caps = [P + 1 + (i % 3) for i in range(17)]
    res = {}
    for mode in ("merged", "step"):
        if mode == "step":
            monkeypatch.setenv("WSEG_NO_PROMPT_PASS", "1")
        for slots in (17, 4):
            res[mode, slots] = gen(eng, x, nb, 40, window_max_length=caps, n_slots=slots, refill_min=1)
        res[mode, "short"] = gen(eng, x, nb, P + 1, n_slots=6)      # max_length = P + 1: one generated token per window
    for key in ((17), (4), ("short")):
        assert torch.equal(res["merged", key][0], res["step", key][0]) and torch.equal(res["merged", key][1], res["step", key][1]), key
    assert all(int(v) <= c for v, c in zip(res["merged", 17][1].tolist(), caps))
    assert (res["merged", "short"][1] == P + 1).all()

"""Error behaviour of the C-ABI (include/wseg.h): bad arguments are rejected with a status and a message, never by a crash.
The model-object half needs no GPU (wseg_model_create / set_tensor / ready / workspace_bytes only fill host structures); the
generate half runs on the device."""
import ctypes as C
import json
import os

import pytest
import torch

from conftest import GOLDEN

INVALID, STATE = -1, -3


def cfg_tiny(**over):
    from whisperseg_amd import _lib
    kw = dict(d_model=128, n_heads=2, enc_layers=2, dec_layers=2, ffn=512, vocab=1280, n_mels=80, spec_cols=1000,
              enc_positions=500, dec_positions=448, dtype=1)
    kw.update(over)
    return _lib.ModelConfig(**kw)


def err(lib):
    return lib.wseg_last_error().decode()


def test_model_create_rejects_bad_geometry():
    from whisperseg_amd import _lib
    lib = _lib.load()
    h = C.c_void_p()
    for over, word in ((dict(d_model=100), "d_model"), (dict(n_heads=3), "d_model"), (dict(ffn=500), "multiples"),
                       (dict(spec_cols=999), "spec_cols"), (dict(enc_positions=64, spec_cols=128), "spec_cols"),
                       (dict(n_mels=200), "n_mels"), (dict(dec_positions=1000), "dec_positions"), (dict(dtype=7), "dtype"),
                       (dict(enc_layers=0), "layer")):
        cfg = cfg_tiny(**over)
        assert lib.wseg_model_create(C.byref(cfg), C.byref(h)) == INVALID, over
        assert word in err(lib), (over, err(lib))
    assert lib.wseg_model_create(None, C.byref(h)) == INVALID


def test_tensor_attachment_errors():
    from whisperseg_amd import _lib
    lib = _lib.load()
    h = C.c_void_p()
    cfg = cfg_tiny()
    assert lib.wseg_model_create(C.byref(cfg), C.byref(h)) == 0
    try:
        assert lib.wseg_model_ready(h) == STATE and "has not been attached" in err(lib)
        buf = (C.c_char * 4096)()
        base = (C.addressof(buf) + 63) & ~63
        assert lib.wseg_model_set_tensor(h, b"enc.nope", base, 256) == INVALID and "unknown tensor" in err(lib)
        assert lib.wseg_model_set_tensor(h, b"enc.ln.g", base, 100) == INVALID and "expected 256 bytes" in err(lib)
        assert lib.wseg_model_set_tensor(h, b"enc.ln.g", base + 2, 256) == INVALID and "aligned" in err(lib)
        assert lib.wseg_model_set_tensor(h, b"enc.ln.g", base, 256) == 0
        assert lib.wseg_model_set_tensor(h, None, base, 256) == INVALID
        # workspace queries: zero for bad requests, monotone in slots and positions otherwise
        assert lib.wseg_workspace_bytes(h, 0, 4, 448) == 0 and lib.wseg_workspace_bytes(h, 4, 0, 448) == 0
        assert lib.wseg_workspace_bytes(h, 4, 9, 448) == 0 and lib.wseg_workspace_bytes(h, 4, 4, 0) == 0
        a, b, c = (lib.wseg_workspace_bytes(h, s, 4, l) for s, l in ((4, 64), (8, 64), (8, 448)))
        assert 0 < a < b < c
        st = _lib.GenerateStats()
        assert lib.wseg_last_stats(h, C.byref(st)) == STATE
        t = (C.c_float * 4)()
        assert lib.wseg_last_timing(h, C.byref(t)) == STATE
    finally:
        lib.wseg_model_destroy(h)
    lib.wseg_model_destroy(None)           # a null handle is ignored


@pytest.mark.gpu
def test_generate_rejects_bad_requests(gpu_lib):
    from safetensors.torch import load_file
    from whisperseg_amd import _lib
    from whisperseg_amd.engine import Engine
    model_dir = os.path.join(GOLDEN, "tiny_model")
    sd = {k: v.float() for k, v in load_file(os.path.join(model_dir, "model.safetensors")).items()}
    with open(os.path.join(model_dir, "config.json")) as f:
        eng = Engine.from_state_dict(sd, json.load(f), "cuda:0", "f32")
    lib = eng.lib
    W, L = 3, 32
    feats = torch.zeros(W, 80, 1000, device="cuda")
    ws = torch.empty(lib.wseg_workspace_bytes(eng.handle, W, 4, L), dtype=torch.uint8, device="cuda")
    toks = torch.full((W, L), -7, dtype=torch.int32, device="cuda")
    lens = torch.full((W,), -7, dtype=torch.int32, device="cuda")

    def call(n_windows=W, ws_bytes=None, feats_ptr=None, **over):
        gp = _lib.GenerateParams()
        gp.prompt[0], gp.prompt[1], gp.prompt[2] = 1, 2, 3
        gp.prompt_len, gp.eos_token_id, gp.pad_token_id, gp.max_length, gp.num_beams, gp.length_penalty = 3, 0, 0, L, 4, 1.0
        for k, v in over.items():
            setattr(gp, k, v)
        return lib.wseg_generate(eng.handle, feats.data_ptr() if feats_ptr is None else feats_ptr, n_windows, C.byref(gp),
                                 ws.data_ptr(), ws.numel() if ws_bytes is None else ws_bytes, toks.data_ptr(), lens.data_ptr(),
                                 _lib.stream_ptr())
    for over, word in ((dict(num_beams=0), "num_beams"), (dict(num_beams=9), "num_beams"), (dict(prompt_len=0), "prompt_len"),
                       (dict(prompt_len=9), "prompt_len"), (dict(max_length=3), "max_length"), (dict(max_length=449), "max_length"),
                       (dict(n_suppress=2), "suppress"), (dict(n_begin_suppress=-1), "suppress"), (dict(n_slots=-1), "scheduler"),
                       (dict(refill_min=-1), "scheduler"), (dict(lookahead=-2), "scheduler"),
                       (dict(num_beams=1, top_k=17), "top_k")):
        assert call(**over) == INVALID, over
        assert word in err(lib), (over, err(lib))
    assert call(ws_bytes=1 << 20) == STATE and "workspace too small" in err(lib)
    assert call(feats_ptr=0) == INVALID and "null" in err(lib)
    assert lib.wseg_generate(None, feats.data_ptr(), W, None, ws.data_ptr(), ws.numel(), toks.data_ptr(), lens.data_ptr(), None) == INVALID
    # zero windows: nothing to do, nothing touched
    assert call(n_windows=0) == 0
    torch.cuda.synchronize()
    assert int(toks.min()) == -7 and int(lens.min()) == -7
    # and the same handle still works afterwards
    assert call() == 0
    torch.cuda.synchronize()
    assert int(lens.min()) >= 4 and int(lens.max()) <= L
    st = _lib.GenerateStats()
    assert lib.wseg_last_stats(eng.handle, C.byref(st)) == 0 and st.n_windows == W and st.n_slots == W
    # encode: zero windows is a no-op, null output is rejected
    assert lib.wseg_encode(eng.handle, feats.data_ptr(), 0, ws.data_ptr(), ws.numel(), toks.data_ptr(), _lib.stream_ptr()) == 0
    assert lib.wseg_encode(eng.handle, feats.data_ptr(), W, ws.data_ptr(), ws.numel(), None, _lib.stream_ptr()) == INVALID


@pytest.mark.gpu
def test_only_an_allocation_failure_halves_the_slot_count(gpu_lib, monkeypatch):
    """Engine.generate answers ONLY the allocation failure of the decode workspace (WsegOutOfMemory) with fewer slots — with a
    warning, recorded in last_stats() — and never when the caller pinned the count (n_slots / $WSEG_SLOTS), never in the plain 16-bit
    modes (their tokens follow the row count), never for another error (a rejected request used to be retried down to 1 slot)."""
    import warnings
    from whisperseg_amd import _lib
    from whisperseg_amd.engine import Engine
    from tools import tiny_model as TM
    mdir = os.path.join(GOLDEN, "tiny_model")
    x = torch.randn(12, 80, 1000, device="cuda") * 0.3
    kw = dict(max_length=12, num_beams=2, suppress_tokens=TM.SUPPRESS, begin_suppress_tokens=TM.BEGIN_SUPPRESS)

    def failing(engine, limit, exc):
        real = engine._workspace

        def ws(n_slots, *a, **k):
            if n_slots > limit:
                raise exc
            return real(n_slots, *a, **k)
        return ws

    eng = Engine.from_pretrained(mdir, "cuda:0", "f16m6")
    want, want_l = eng.generate(x, TM.PROMPT, TM.EOT, TM.EOT, **kw)
    monkeypatch.delenv("WSEG_SLOTS", raising=False)
    monkeypatch.setattr(eng, "_workspace", failing(eng, 3, _lib.WsegOutOfMemory("cannot allocate")))
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        got, got_l = eng.generate(x, TM.PROMPT, TM.EOT, TM.EOT, **kw)
    st = eng.last_stats()
    assert st["n_slots"] == 3 and st["slots_halved"] == 2                      # 12 -> 6 -> 3
    assert len(caught) == 2 and "near-tie" in str(caught[0].message)
    assert torch.equal(got, want) and torch.equal(got_l, want_l)                # the trained model: tokens independent of the slot count
    with pytest.raises(_lib.WsegOutOfMemory):                                   # pinned by the caller
        eng.generate(x, TM.PROMPT, TM.EOT, TM.EOT, n_slots=12, **kw)
    monkeypatch.setenv("WSEG_SLOTS", "12")
    with pytest.raises(_lib.WsegOutOfMemory):                                   # pinned through the environment
        eng.generate(x, TM.PROMPT, TM.EOT, TM.EOT, **kw)
    monkeypatch.delenv("WSEG_SLOTS")
    monkeypatch.setattr(eng, "_workspace", failing(eng, 3, _lib.WsegError("wseg_workspace_bytes rejected the request")))
    with pytest.raises(_lib.WsegError, match="rejected"):                       # not an allocation failure: no retry
        eng.generate(x, TM.PROMPT, TM.EOT, TM.EOT, **kw)
    plain = Engine.from_pretrained(mdir, "cuda:0", "bf16")
    monkeypatch.setattr(plain, "_workspace", failing(plain, 3, _lib.WsegOutOfMemory("cannot allocate")))
    with pytest.raises(_lib.WsegOutOfMemory):                                   # plain 16-bit mode: the error stands
        plain.generate(x, TM.PROMPT, TM.EOT, TM.EOT, **kw)

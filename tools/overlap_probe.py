"""Can the encoder of the NEXT admission run beside the decode steps of the CURRENT slots?  

Two engines over the same weights on one GPU (whisperseg-large geometry, default mode): engine E encodes 256-window passes (MFMA-bound
GEMMs: the persistent 256x256 kernel holds 8 waves x 256 registers + 128 KB of LDS per CU), engine D decodes 1 024 slots from
precomputed encoder states (cross-attention at the HBM roof + latency-bound GEMM / reduction launches).  Measured, each on its own stream:

  1. alone:       T_enc (n passes), T_dec (one 34-step call)
  2. together:    both submitted at once from two host threads -> wall time, against T_enc + T_dec (no overlap) and max(T_enc, T_dec)
  3. CU masks:    the decode call on a stream restricted to 32 / 64 / 128 / 256 CUs (hipExtStreamCreateWithCUMask): how much of the chip an
                  HBM-bound step needs; and the encoder passes on the complementary mask beside it.

    python tools/overlap_probe.py [--passes 4] [--windows 1024]        -> profiles/r05_overlap.txt (by hand)
"""
import argparse
import ctypes
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from whisperseg_amd.engine import Engine  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--passes", type=int, default=4)
ap.add_argument("--windows", type=int, default=1024)
ap.add_argument("--dtype", default="f16m6")
a = ap.parse_args()
cfg = dict(d_model=1280, encoder_attention_heads=20, decoder_attention_heads=20, encoder_layers=32, decoder_layers=32, encoder_ffn_dim=5120,
           decoder_ffn_dim=5120, vocab_size=51865, num_mel_bins=80, max_source_positions=500, max_target_positions=448)
D = Engine.random(cfg, "cuda:0", a.dtype, seed=0)
E = D.sibling(a.dtype)
feats = torch.randn(256, 80, 1000, device="cuda") * 0.5
enc_states = torch.cat([D.encode(feats[:64]) for _ in range(a.windows // 64)])
prompt, eos = [50258, 50259, 50363], 50257
kw = dict(max_length=35, num_beams=4, suppress_tokens=[eos, 1, 2], begin_suppress_tokens=[220], n_slots=a.windows)
dummy = torch.zeros(a.windows, 80, 1000, device="cuda")


def run_dec(stream):
    with torch.cuda.stream(stream):
        D.generate(dummy, prompt, eos, eos, encoder_output=enc_states, **kw)


def run_enc(stream, n=a.passes):
    with torch.cuda.stream(stream):
        for _ in range(n):
            E.encode(feats)


def timed(fns):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=f) for f in fns]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
for _ in range(2):      # warm-up: workspaces, step graph
    run_dec(sb); run_enc(sa, 1)
torch.cuda.synchronize()
t_enc = min(timed([lambda: run_enc(sa)]) for _ in range(2))
t_dec = min(timed([lambda: run_dec(sb)]) for _ in range(2))
t_both = min(timed([lambda: run_enc(sa), lambda: run_dec(sb)]) for _ in range(2))
print(f"{a.dtype}, {a.windows} slots x 34 decode steps beside {a.passes} encoder passes of 256 windows")
print(f"alone:    encoder passes {t_enc:8.1f} ms   decode call {t_dec:8.1f} ms   sum {t_enc + t_dec:8.1f}   max {max(t_enc, t_dec):8.1f}")
print(f"together: {t_both:8.1f} ms  = {t_both / (t_enc + t_dec):.3f} of the sum (1.0 = no overlap at all; {max(t_enc, t_dec) / (t_enc + t_dec):.3f} = perfect overlap)")

# ---- CU masks ----
hip = None
for name in ("libamdhip64.so", "libamdhip64.so.6", "libamdhip64.so.7"):
    try:
        hip = ctypes.CDLL(name)
        break
    except OSError:
        continue
if hip is None or not hasattr(hip, "hipExtStreamCreateWithCUMask"):
    print("hipExtStreamCreateWithCUMask not available: CU-mask part skipped")
    sys.exit(0)


def masked_stream(cus):      # `cus` of the 256 CUs, spread evenly over the 8 XCDs (CU c of the mask = bit c; XCD = c % 8 in enumeration order
    bits = 0                 # is not documented: the mask keeps every `256 // cus`-th CU, which takes the same share of every XCD either way)
    step = 256 // cus
    for c in range(0, 256, step):
        bits |= 1 << c
    words = (ctypes.c_uint32 * 8)(*[(bits >> (32 * i)) & 0xFFFFFFFF for i in range(8)])
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value), bits


def complement_stream(bits):
    inv = ((1 << 256) - 1) ^ bits
    words = (ctypes.c_uint32 * 8)(*[(inv >> (32 * i)) & 0xFFFFFFFF for i in range(8)])
    st = ctypes.c_void_p()
    assert hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words) == 0
    return torch.cuda.ExternalStream(st.value)


for cus in (256, 128, 64, 32):
    sd, bits = masked_stream(cus)
    run_dec(sd); torch.cuda.synchronize()      # the step graph is captured per stream
    td = min(timed([lambda: run_dec(sd)]) for _ in range(2))
    line = f"decode call on {cus:3d} CUs: {td:8.1f} ms ({td / t_dec:.2f}x the whole chip)"
    if cus < 256:
        se = complement_stream(bits)
        run_enc(se, 1); torch.cuda.synchronize()
        te = min(timed([lambda: run_enc(se)]) for _ in range(2))
        tb = min(timed([lambda: run_enc(se), lambda: run_dec(sd)]) for _ in range(2))
        line += f" | encoder passes on the other {256 - cus:3d} CUs: alone {te:8.1f} ms ({te / t_enc:.2f}x), both at once {tb:8.1f} ms = {tb / (t_enc + t_dec):.3f} of the unpartitioned sum"
    print(line, flush=True)

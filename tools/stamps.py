"""Where does a decode kernel's time go?  In-kernel s_memrealtime stamps (100 MHz) at the phase boundaries of one workgroup:
    python -m whisperseg_amd.build --stamps N        (here: N = 1 self-attention, 2 packed cross-attention, 3 24-bit cross-attention)
    WSEG_LIB=whisperseg_amd/lib/libwseg_stamps<N>.so python tools/stamps.py N [--windows 8] [--dtype bf16|f16x3]     (GPU box)
Phases: 0 entry, 1 idle-slot check, 2 query (q | k | v) reduced from the split-K partials, 3 scores, 4 barrier, 5 softmax, 6 P V, 7 end."""
import argparse, ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from whisperseg_amd import _lib
from whisperseg_amd.engine import Engine

ap = argparse.ArgumentParser()
ap.add_argument("kernel", type=int)
ap.add_argument("--windows", type=int, default=8)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--layers", type=int, default=32)
a = ap.parse_args()
cfg = dict(d_model=1280, encoder_attention_heads=20, decoder_attention_heads=20, encoder_layers=2, decoder_layers=a.layers,
           encoder_ffn_dim=5120, decoder_ffn_dim=5120, vocab_size=51865, num_mel_bins=80, max_source_positions=500, max_target_positions=448)
eng = Engine.random(cfg, "cuda:0", a.dtype)
feats = torch.randn(a.windows, 80, 1000, device="cuda") * 0.5
lib = ctypes.CDLL(_lib.LIB_PATH)
names = ["entry", "idle check", "q reduced", "scores", "barrier", "softmax", "P V", "end"]
for it in range(3):
    eng.generate(feats, [50258, 50259, 50363], 50257, 50257, max_length=20, num_beams=4, suppress_tokens=[50257, 1, 2], begin_suppress_tokens=[220],
                 n_slots=a.windows)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 32)()
    lib.wseg_debug_stamps(buf)
    t = [buf[i] for i in range(8)]
    print(f"kernel {a.kernel}, {a.windows} slots, {a.dtype}: " + "  ".join(f"{n} {(x - t[0]) / 100.0:.2f}" for n, x in zip(names, t)) + "  (us)", flush=True)

"""First-step logit error and token agreement of engine modes against the exact-parity f32 mode on the SAME fp32 weights at
whisperseg-large geometry (32 + 32 layers, seeded random weights; needs the GPU):

    python tools/logit_error.py [--windows 4] [--gen 32] [--layers 32] MODE [MODE ...]      e.g. bf16x3 f16x3 f16 bf16

Prints one JSON line per mode: max |logit diff|, logit scale, worst row cosine, first-token and whole-sequence agreement."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisperseg_amd.engine import DTYPES, Engine, geometry_from_config, random_weights, to_engine_layout  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--windows", type=int, default=4)
ap.add_argument("--gen", type=int, default=32)
ap.add_argument("--layers", type=int, default=32)
ap.add_argument("--beams", type=int, default=4)
ap.add_argument("modes", nargs="+")
a = ap.parse_args()
cfg = dict(d_model=1280, encoder_attention_heads=20, decoder_attention_heads=20, encoder_layers=a.layers, decoder_layers=a.layers,
           encoder_ffn_dim=5120, decoder_ffn_dim=5120, vocab_size=51865, num_mel_bins=80, max_source_positions=500, max_target_positions=448)
geo = geometry_from_config(cfg)
w32 = random_weights(geo, torch.float32, "cuda:0", seed=0)
feats = torch.randn(a.windows, 80, 1000, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1)) * 0.5
prompt, eos = [50258, 50259, 50363], 50257
kw = dict(max_length=3 + a.gen, num_beams=a.beams, suppress_tokens=[eos, 1, 2], begin_suppress_tokens=[220], return_first_logits=True)
ref_eng = Engine(geo, w32, "cuda:0", "f32")
rt, rl, ref = ref_eng.generate(feats, prompt, eos, eos, **kw)
ref_enc = ref_eng.encode(feats).float()
rt, rl, ref = rt.cpu(), rl.cpu(), ref.float().cpu()
del ref_eng
for mode in a.modes:
    eng = Engine(geo, to_engine_layout(w32, mode), "cuda:0", mode)
    t, l, fl = eng.generate(feats, prompt, eos, eos, **kw)
    enc_err = (eng.encode(feats).float() - ref_enc).abs().max().item()
    t, l, fl = t.cpu(), l.cpu(), fl.float().cpu()
    same = [bool(l[i] == rl[i] and torch.equal(t[i, :l[i]], rt[i, :rl[i]])) for i in range(a.windows)]
    first = [bool(t[i, 3] == rt[i, 3]) for i in range(a.windows)]
    print(json.dumps(dict(mode=mode, layers=a.layers, windows=a.windows, max_abs_logit_err=(fl - ref).abs().max().item(),
                          logit_scale=ref.abs().max().item(), cosine_min=torch.nn.functional.cosine_similarity(fl, ref, dim=1).min().item(),
                          encoder_out_max_abs_err=enc_err, encoder_out_scale=ref_enc.abs().max().item(),
                          first_token_equal=f"{sum(first)}/{a.windows}", sequences_equal=f"{sum(same)}/{a.windows}")), flush=True)
    del eng
    torch.cuda.empty_cache()

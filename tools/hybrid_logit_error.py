"""Attribution of a mode's first-step logit error to its ENCODER and its DECODER at whisperseg-large geometry (32 + 32 layers, seeded random
fp32 weights; needs the GPU): the encoder states of mode A are handed to the decoder of mode B (wseg_generate's encoder_output hook) and the
first-step logits are compared with the exact-parity f32 mode.

    python tools/hybrid_logit_error.py [--windows 8] [--layers 32]

r06: asked whether a hybrid (encoder on the fp6 MX path, decoder on three MFMAs) could keep f16x3's decisions at part of f16m6's speed."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from whisperseg_amd.engine import Engine, geometry_from_config, random_weights, to_engine_layout  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--windows", type=int, default=8)
ap.add_argument("--layers", type=int, default=32)
a = ap.parse_args()
cfg = dict(d_model=1280, encoder_attention_heads=20, decoder_attention_heads=20, encoder_layers=a.layers, decoder_layers=a.layers,
           encoder_ffn_dim=5120, decoder_ffn_dim=5120, vocab_size=51865, num_mel_bins=80, max_source_positions=500, max_target_positions=448)
geo = geometry_from_config(cfg)
w32 = random_weights(geo, torch.float32, "cuda:0", seed=0)
feats = torch.randn(a.windows, 80, 1000, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1)) * 0.5
prompt, eos = [50258, 50259, 50363], 50257
kw = dict(max_length=3 + 8, num_beams=4, suppress_tokens=[eos, 1, 2], begin_suppress_tokens=[220], return_first_logits=True)
engines = {m: Engine(geo, to_engine_layout(w32, m), "cuda:0", m) for m in ("f32", "f16x3", "f16m6")}
ref = engines["f32"].generate(feats, prompt, eos, eos, **kw)[2].float().cpu()
enc = {m: e.encode(feats).float() for m, e in engines.items()}
for em in ("f32", "f16x3", "f16m6"):
    for dm in ("f32", "f16x3", "f16m6"):
        fl = engines[dm].generate(feats, prompt, eos, eos, encoder_output=enc[em], **kw)[2].float().cpu()
        print(json.dumps(dict(encoder=em, decoder=dm, max_abs_logit_err=(fl - ref).abs().max().item(), mean_abs_logit_err=(fl - ref).abs().mean().item(),
                              logit_scale=ref.abs().max().item())), flush=True)

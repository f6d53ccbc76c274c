"""Shared definitions for the tiny fixture model (tools/train_tiny.py, tools/make_golden.py).

A WhisperSeg-geometry model small enough to commit (≈1.4 M parameters stored as bf16):
80 mel bins, 1000 spectrogram columns, 500 encoder positions, head_dim 64, the reference's
label grammar (reference datautils.py:354-368) over a compact 1280-entry vocabulary whose
first 256 ids follow the GPT-2 byte layout of the real Whisper vocabulary (digits '0'..'9'
are ids 15..24, as in reference SURVEY §8 a-8).
"""
import json
import os

import numpy as np

VOCAB_SIZE = 1280
EOT = 256
SOT = 257
EN = 258
NOTIMESTAMPS = 259
TIME0 = 260                      # <|0|> .. <|1000|>  -> 260 .. 1260
SPECIES = ["<|zebra_finch|>", "<|bengalese_finch|>", "<|mouse|>", "<|marmoset|>", "<|human|>",
           "<|unknown|>", "<|animal|>"]
SPECIES0 = 1261
PROMPT = [SOT, EN, NOTIMESTAMPS]
SUPPRESS = [1, 2, 7, 8, 9, 10, 14, 25, 26, 27, 28, 29, 31, 58, 59, 60, 61, 62, 63, 90, 91, 92, 93,
            359, 503, 522, 542, 873, 893, 902, 918, 922, 931, SOT]
BEGIN_SUPPRESS = [220, EOT]
CLUSTER_CODEBOOK = {"a": 0, "b": 1, "c": 2}
SR = 16000
STS = 0.01
TONES = [500.0, 1500.0, 3000.0]


def bytes_to_unicode():
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return dict(zip(bs, [chr(c) for c in cs]))


def added_tokens():
    d = {"<|endoftext|>": EOT, "<|startoftranscript|>": SOT, "<|en|>": EN, "<|notimestamps|>": NOTIMESTAMPS}
    for i in range(1001):
        d["<|%d|>" % i] = TIME0 + i
    for i, s in enumerate(SPECIES):
        d[s] = SPECIES0 + i
    return d


def base_vocab():
    b2u = bytes_to_unicode()
    order = list(b2u.keys())          # the GPT-2 ordering: '!'..'~', '¡'..'¬', '®'..'ÿ', then the rest
    return {b2u[b]: i for i, b in enumerate(order)}


def hf_config_dict():
    return dict(
        model_type="whisper", vocab_size=VOCAB_SIZE, num_mel_bins=80, d_model=128,
        encoder_layers=2, decoder_layers=2, encoder_attention_heads=2, decoder_attention_heads=2,
        encoder_ffn_dim=512, decoder_ffn_dim=512, max_source_positions=500, max_target_positions=448,
        decoder_start_token_id=SOT, pad_token_id=EOT, eos_token_id=EOT, bos_token_id=EOT,
        activation_function="gelu", scale_embedding=False,
        total_spec_columns=1000, cluster_codebook=CLUSTER_CODEBOOK,
        default_segmentation_config={"sr": SR, "spec_time_step": STS, "min_frequency": 0},
    )


def write_model_dir(path, state_dict_bf16):
    """HF-style directory: config.json, generation_config.json, vocab.json, added_tokens.json, model.safetensors."""
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(hf_config_dict(), f, indent=1)
    with open(os.path.join(path, "generation_config.json"), "w") as f:
        json.dump({"max_length": 448, "suppress_tokens": SUPPRESS, "begin_suppress_tokens": BEGIN_SUPPRESS,
                   "pad_token_id": EOT, "eos_token_id": EOT, "decoder_start_token_id": SOT}, f, indent=1)
    with open(os.path.join(path, "vocab.json"), "w") as f:
        json.dump(base_vocab(), f, ensure_ascii=False)
    with open(os.path.join(path, "added_tokens.json"), "w") as f:
        json.dump(added_tokens(), f)
    save_file({k: v.contiguous() for k, v in state_dict_bf16.items()}, os.path.join(path, "model.safetensors"))


def synth_clip(rng, n_samples=160000, max_events=9):
    """Noise + tone bursts; returns (float32 audio, [(onset_s, offset_s, cluster_id)])."""
    t = np.arange(n_samples) / SR
    x = 0.005 * rng.standard_normal(n_samples)
    events = []
    cur = rng.uniform(0.0, 1.5)
    n_ev = rng.integers(0, max_events + 1)
    for _ in range(n_ev):
        dur = rng.uniform(0.2, 1.0)
        if cur + dur > n_samples / SR - 0.05:
            break
        c = int(rng.integers(0, 3))
        amp = rng.uniform(0.1, 0.3)
        m = (t >= cur) & (t < cur + dur)
        x[m] += amp * np.sin(2 * np.pi * TONES[c] * t[m])
        events.append((cur, cur + dur, c))
        cur += dur + rng.uniform(0.12, 1.2)
    return x.astype(np.float32), events


def label_tokens(events, species="<|unknown|>"):
    """reference datautils.py:354-368 grammar -> token ids (after the 3-token prompt), EOT-terminated."""
    ids = [SPECIES0 + SPECIES.index(species)]
    for on, off, c in events:
        ids += [TIME0 + int(np.round(on / STS / 2)), 15 + c, TIME0 + int(np.round(off / STS / 2))]
    return ids + [EOT]

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

# Fixture-model variants.  "tiny" is the model of rounds 1-5 (tests/golden/tiny_model: weights rounded to bf16, the 200-recording
# sweep every precision format of r03-r05 was chosen on).  "tiny2" (r06) is the HELD-OUT model: another shape (d 256, 4 heads,
# 3 + 3 layers, ffn 768), another init seed, another data stream, other tone frequencies / amplitudes / noise floor, and
# FULL-MANTISSA fp32 weights (a real checkpoint's weights are not representable in 16 bits: with the bf16-rounded weights of "tiny"
# the lo halves of every weight operand are zero in the IEEE-half modes).
VARIANTS = {
    "tiny": dict(d_model=128, heads=2, enc_layers=2, dec_layers=2, ffn=512, tones=TONES, amp=(0.1, 0.3), noise=0.005,
                 init_seed=0, data_seed=1234, round_bf16=True),
    "tiny2": dict(d_model=256, heads=4, enc_layers=3, dec_layers=3, ffn=768, tones=[700.0, 2100.0, 4200.0], amp=(0.05, 0.35),
                  noise=0.008, init_seed=1, data_seed=987654, round_bf16=False),
    # "tiny3" (r06): a THIRD model — deeper and narrower (d 128, 2 heads, 4 + 4 layers, ffn 640), its own seeds and signal family, fp32
    # weights — for one more held-out sweep on another architecture (tests/golden/tiny_model3, tiny3_sweep.json)
    "tiny3": dict(d_model=128, heads=2, enc_layers=4, dec_layers=4, ffn=640, tones=[600.0, 1800.0, 3600.0], amp=(0.08, 0.3),
                  noise=0.006, init_seed=2, data_seed=424242, round_bf16=False),
}


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


def hf_config_dict(variant="tiny"):
    v = VARIANTS[variant]
    return dict(
        model_type="whisper", vocab_size=VOCAB_SIZE, num_mel_bins=80, d_model=v["d_model"],
        encoder_layers=v["enc_layers"], decoder_layers=v["dec_layers"], encoder_attention_heads=v["heads"],
        decoder_attention_heads=v["heads"],
        encoder_ffn_dim=v["ffn"], decoder_ffn_dim=v["ffn"], max_source_positions=500, max_target_positions=448,
        decoder_start_token_id=SOT, pad_token_id=EOT, eos_token_id=EOT, bos_token_id=EOT,
        activation_function="gelu", scale_embedding=False,
        total_spec_columns=1000, cluster_codebook=CLUSTER_CODEBOOK,
        default_segmentation_config={"sr": SR, "spec_time_step": STS, "min_frequency": 0},
    )


def write_model_dir(path, state_dict_bf16, variant="tiny"):
    """HF-style directory: config.json, generation_config.json, vocab.json, added_tokens.json, model.safetensors."""
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(hf_config_dict(variant), f, indent=1)
    with open(os.path.join(path, "generation_config.json"), "w") as f:
        json.dump({"max_length": 448, "suppress_tokens": SUPPRESS, "begin_suppress_tokens": BEGIN_SUPPRESS,
                   "pad_token_id": EOT, "eos_token_id": EOT, "decoder_start_token_id": SOT}, f, indent=1)
    with open(os.path.join(path, "vocab.json"), "w") as f:
        json.dump(base_vocab(), f, ensure_ascii=False)
    with open(os.path.join(path, "added_tokens.json"), "w") as f:
        json.dump(added_tokens(), f)
    save_file({k: v.contiguous() for k, v in state_dict_bf16.items()}, os.path.join(path, "model.safetensors"))


def synth_clip(rng, n_samples=160000, max_events=9, variant="tiny"):
    """Noise + tone bursts; returns (float32 audio, [(onset_s, offset_s, cluster_id)])."""
    v = VARIANTS[variant]
    tones, amp_lo_hi, noise = v["tones"], v["amp"], v["noise"]
    t = np.arange(n_samples) / SR
    x = noise * rng.standard_normal(n_samples)
    events = []
    cur = rng.uniform(0.0, 1.5)
    n_ev = rng.integers(0, max_events + 1)
    for _ in range(n_ev):
        dur = rng.uniform(0.2, 1.0)
        if cur + dur > n_samples / SR - 0.05:
            break
        c = int(rng.integers(0, 3))
        amp = rng.uniform(*amp_lo_hi)
        m = (t >= cur) & (t < cur + dur)
        x[m] += amp * np.sin(2 * np.pi * tones[c] * t[m])
        events.append((cur, cur + dur, c))
        cur += dur + rng.uniform(0.12, 1.2)
    return x.astype(np.float32), events


def label_tokens(events, species="<|unknown|>"):
    """reference datautils.py:354-368 grammar -> token ids (after the 3-token prompt), EOT-terminated."""
    ids = [SPECIES0 + SPECIES.index(species)]
    for on, off, c in events:
        ids += [TIME0 + int(np.round(on / STS / 2)), 15 + c, TIME0 + int(np.round(off / STS / 2))]
    return ids + [EOT]

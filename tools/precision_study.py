"""CPU study: which tensors of the Whisper forward need more than 16 bits for the north-star tolerance?

Runs the 200-recording parity sweep (tests/golden/tiny_sweep.json, rows recorded from the reference) through the CPU oracle's
decoding loop (oracle/whisper_ref.py: HF greedy / beam semantics) with the model arithmetic EMULATING a storage / operand
precision policy per operator class, and scores the rows exactly as tools/parity_sweep.py scores the GPU engine.  Test / design
infrastructure only (imports oracle/); nothing here ships.

    python tools/precision_study.py out.json POLICY [POLICY ...]       (CPU, ~1-2 min per policy)

POLICY = comma-separated  class=fmt  pairs, classes:
    gemm   operands of every Linear / conv (activations and weights)          fmt: f32 | f16 | bf16 | bf16x3 | f16x3
    eattn  encoder attention tensors Q, K, V, P (softmax probabilities)        fmt: f32 | f16 | bf16 | bf16x2 | f16x2
    ckv    cross-attention K / V storage (ck / cv: one of them)                (x2 = hi + lo pair, i.e. ~16 / ~22 mantissa bits)
    skv    decoder self-attention K / V cache
    dq     decoder attention queries (self + cross) and attention outputs are GEMM operands -> class gemm; dq = the query
    egemm / dgemm / ckvg / lm   per-STAGE override of `gemm`: encoder Linears + convs / decoder-layer Linears / the cross-K|V
           projections / the LM head.  Extra formats for the GEMM classes: f16x2w | bf16x2w (weights hi + lo, activations rounded
           once: 2 MFMAs per product) and f16x2a | bf16x2a (activations hi + lo, weights rounded once)
    e.qkv / e.o / e.fc1 / e.fc2 / e.conv / d.qkv / d.o / d.cq / d.co / d.fc1 / d.fc2   per-OPERATOR override inside a stage
           (round 4: which GEMMs of the encoder / decoder layers carry the need for split operands?)
    all    shorthand: every class
e.g.  "all=f16"   "gemm=bf16x3"   "gemm=bf16x3,ckv=f16"   "gemm=bf16x3,eattn=f16,ckv=f16,skv=f16"
The special policy "margins" dumps the histogram of top-1 / top-2 logit margins of the fp32 oracle over the sweep.
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_inputs as GI  # noqa: E402
from oracle import frontend as FE  # noqa: E402
from oracle import whisper_ref as W  # noqa: E402
from tools import tiny_model as TM  # noqa: E402
from whisperseg_amd import postprocess  # noqa: E402
from whisperseg_amd.model import SegmenterBase  # noqa: E402
from whisperseg_amd.tokenizer import WhisperSegTokenizer  # noqa: E402

# STUDY_SWEEP=sweep2_heldout (r06): the held-out 1 000-recording sweep of the second fixture model instead of the 200 recordings every
# format was chosen on (tools/parity_sweep.py SWEEPS)
from tools.parity_sweep import SWEEPS  # noqa: E402
SWEEP_ROWS, SWEEP_MODEL, SWEEP_VARIANT = SWEEPS[os.environ.get("STUDY_SWEEP", "sweep1")]
MODEL_DIR = os.path.join(ROOT, "tests", "golden", SWEEP_MODEL)


def rnd(x, fmt):
    if fmt == "f32":
        return x
    if fmt == "f16":
        return x.half().float()
    if fmt == "bf16":
        return x.bfloat16().float()
    if fmt in ("bf16x2", "bf16x3"):
        hi = x.bfloat16().float()
        return hi + (x - hi).bfloat16().float()
    if fmt in ("f16x2", "f16x3"):
        hi = x.half().float()
        return hi + (x - hi).half().float()
    if fmt in ("t24", "t22", "t20"):      # fp32 rounded (half up in magnitude) to its top 24 / 22 / 20 bits: 16 / 14 / 12-bit significands
        drop = 32 - int(fmt[1:])          # (t24 = the 24-bit cross K / V of the split modes, EpiParams::kv24)
        w = x.contiguous().view(torch.int32) + (1 << (drop - 1))
        return (w & ~((1 << drop) - 1)).view(torch.float32)
    if fmt.startswith("bfp"):             # block floating point along the last dim: bfp<bits>r = one power-of-two scale per ROW (a K / V row of
        bits, blk = fmt[3:].split("r") if "r" in fmt else fmt[3:].split("b")      # one head: 64 elements), bfp<bits>b<n> = per block of n elements;
        bits = int(bits)                                                             # elements are <bits>-bit two's-complement integers
        n = x.shape[-1] if not blk else int(blk)
        xb = x.reshape(*x.shape[:-1], -1, n)
        amax = xb.abs().amax(dim=-1, keepdim=True).clamp_min(1e-38)
        scale = torch.exp2(torch.ceil(torch.log2(amax)) - (bits - 1))              # amax / scale <= 2^(bits-1)
        q = torch.round(xb / scale).clamp(-(2 ** (bits - 1)) + 1, 2 ** (bits - 1) - 1)
        return (q * scale).reshape(x.shape)
    raise ValueError(fmt)


STAGES = ("egemm", "dgemm", "ckvg", "lm")


def e5m2_quant(x, pre_scale=1.0):
    """UNSCALED bf8 (e5m2: 2 mantissa bits, normals down to 2^-14, subnormal step 2^-16, max 57344) of x * pre_scale, returned at
    the original magnitude: no block scales at all — the 5-bit exponent covers the range by itself."""
    y = (x * pre_scale).clamp(-57344.0, 57344.0)
    e = torch.floor(torch.log2(y.abs().clamp_min(1e-38))).clamp(min=-14, max=15)
    step = torch.exp2(e - 2)
    return torch.round(y / step) * step / pre_scale


def mx_quant(x, fmt, block=32):
    """Block-scaled micro-float quantisation along the last dim (blocks of 32, one power-of-two scale per block — what the gfx950
    v_mfma_scale_f32_*_f8f6f4 instructions consume): fmt "e2m3" (fp6: max 7.5, 3 mantissa bits) or "e4m3" (fp8: max 448).  The
    scale is the smallest power of two that brings the block's maximum inside the element range (no saturation), elements are
    rounded to nearest even.  Returns the dequantised tensor."""
    emax, vmax, emin = (2, 7.5, 0) if fmt == "e2m3" else (8, 448.0, -6)
    shp = x.shape
    k = shp[-1]
    pad = (-k) % block
    xp = F.pad(x, (0, pad)) if pad else x
    xb = xp.reshape(*shp[:-1], -1, block)
    amax = xb.abs().amax(dim=-1, keepdim=True).clamp_min(1e-38)
    scale = torch.exp2(torch.ceil(torch.log2(amax / vmax)))
    y = xb / scale
    e = torch.floor(torch.log2(y.abs().clamp_min(1e-38))).clamp(min=emin, max=emax)
    step = torch.exp2(e - 3)
    q = torch.round(y / step) * step                      # torch.round is half-to-even
    out = (q * scale).reshape(*shp[:-1], -1)
    return out[..., :k] if pad else out


class Policy:
    def __init__(self, text):
        self.fmt = dict(gemm="f32", eattn="f32", ck="f32", cv="f32", skv="f32", dq="f32")
        stage = {}
        self.sub = {}
        for part in filter(None, text.split(",")):
            k, v = part.split("=")
            if k[:2] in ("e.", "d."):
                self.sub[k] = v
                continue
            if k in STAGES:
                stage[k] = v
                continue
            for kk in (self.fmt if k == "all" else (["ck", "cv"] if k == "ckv" else [k])):
                self.fmt[kk] = v if not (kk != "gemm" and v.endswith("x3")) else v[:-1] + "2"
        for k in STAGES:
            self.fmt[k] = stage.get(k, self.fmt["gemm"])
        self.wcache = {}

    @staticmethod
    def stage_of(prefix):
        if prefix.startswith("model.encoder."):
            return "egemm"
        if prefix.endswith("encoder_attn.k_proj") or prefix.endswith("encoder_attn.v_proj"):
            return "ckvg"
        if prefix.endswith("embed_tokens"):
            return "lm"
        return "dgemm"

    @staticmethod
    def sub_of(prefix):
        side = "e." if prefix.startswith("model.encoder.") else "d."
        for suffix, name in (("encoder_attn.q_proj", "cq"), ("encoder_attn.out_proj", "co"), ("self_attn.out_proj", "o"),
                             ("self_attn.q_proj", "qkv"), ("self_attn.k_proj", "qkv"), ("self_attn.v_proj", "qkv"), ("fc1", "fc1"), ("fc2", "fc2")):
            if prefix.endswith(suffix):
                return side + name
        return ""

    def linear(self, x, sd, prefix, bias=True):
        fmt = self.sub.get(self.sub_of(prefix)) or self.fmt[self.stage_of(prefix)]
        w = sd[prefix + ".weight"]
        b = sd[prefix + ".bias"] if bias else None
        if fmt == "f32":
            return F.linear(x, w, b)
        if fmt == "f16m8u":
            # cross terms on UNSCALED bf8 (e5m2) copies: hi8 = e5m2(hi), lo8 = e5m2(lo * 2^11) (constant scale, undone by the MFMA's
            # scale operand)
            key = (prefix, fmt)
            if key not in self.wcache:
                wh = w.clamp(-65504, 65504).half().float()
                self.wcache[key] = (wh, e5m2_quant(wh), e5m2_quant(w - wh, 2048.0))
            wh, wh_q, wl_q = self.wcache[key]
            xh = x.clamp(-65504, 65504).half().float()
            y = F.linear(xh, wh) + (F.linear(e5m2_quant(xh), wl_q) + F.linear(e5m2_quant(x - xh, 2048.0), wh_q))
            return y + b if b is not None else y
        if fmt in ("f16m6", "f16m8"):
            # hi x hi on the IEEE-half matrix cores; the two cross terms hi x lo + lo x hi on the block-scaled MX matrix cores
            # (fp6 e2m3 at 4x / fp8 e4m3 at 2x the 16-bit rate): every operand of a cross term is an MX-quantised copy
            ef = "e2m3" if fmt == "f16m6" else "e4m3"
            key = (prefix, fmt)
            if key not in self.wcache:
                wh = w.clamp(-65504, 65504).half().float()
                self.wcache[key] = (wh, mx_quant(wh, ef), mx_quant(w - wh, ef))
            wh, wh_q, wl_q = self.wcache[key]
            xh = x.clamp(-65504, 65504).half().float()
            y = F.linear(xh, wh) + (F.linear(mx_quant(xh, ef), wl_q) + F.linear(mx_quant(x - xh, ef), wh_q))
            return y + b if b is not None else y
        if fmt.endswith("x2w") or fmt.endswith("x2a"):
            base = torch.bfloat16 if fmt.startswith("bf16") else torch.float16
            key = (prefix, fmt)
            if key not in self.wcache:
                wh = w.to(base).float()
                self.wcache[key] = (wh, (w - wh).to(base).float())
            wh, wl = self.wcache[key]
            xh = x.to(base).float()
            if fmt.endswith("x2w"):
                y = F.linear(xh, wh) + F.linear(xh, wl)
            else:
                y = F.linear(xh, wh) + F.linear((x - xh).to(base).float(), wh)
            return y + b if b is not None else y
        if fmt.endswith("x3"):
            base = torch.bfloat16 if fmt.startswith("bf16") else torch.float16
            key = (prefix, fmt)
            if key not in self.wcache:
                wh = w.to(base).float()
                self.wcache[key] = (wh, (w - wh).to(base).float())
            wh, wl = self.wcache[key]
            xh = x.to(base).float()
            xl = (x - xh).to(base).float()
            y = F.linear(xh, wh) + (F.linear(xh, wl) + F.linear(xl, wh))       # lo x lo is dropped, as on the matrix cores
            return y + b if b is not None else y
        return F.linear(rnd(x, fmt), rnd(w, fmt), b)

    def conv(self, x, w, b, **kw):
        fmt = self.sub.get("e.conv") or self.fmt["egemm"]
        if fmt == "f32":
            return F.conv1d(x, w, b, **kw)
        if fmt == "f16m8u":
            wh = w.half().float(); xh = x.half().float()
            return F.conv1d(xh, wh, b, **kw) + (F.conv1d(e5m2_quant(xh), e5m2_quant(w - wh, 2048.0), None, **kw) +
                                                 F.conv1d(e5m2_quant(x - xh, 2048.0), e5m2_quant(wh), None, **kw))
        if fmt in ("f16m6", "f16m8"):
            # as an im2col GEMM: K = tap * C + channel, blocks of 32 along K (the channel dim padded to a multiple of 32 per tap)
            ef = "e2m3" if fmt == "f16m6" else "e4m3"
            o, c, kt = w.shape
            stride, padding = kw.get("stride", 1), kw.get("padding", 0)
            cols = F.unfold(x.unsqueeze(-1), (kt, 1), padding=(padding, 0), stride=(stride, 1))      # [B, C * kt, T']
            bsz, _, tt = cols.shape
            cols = cols.reshape(bsz, c, kt, tt).permute(0, 3, 2, 1)                                   # [B, T', tap, C]
            cp = (-c) % 32
            cols = F.pad(cols, (0, cp)).reshape(bsz, tt, kt * (c + cp))
            wm = F.pad(w.permute(0, 2, 1), (0, cp)).reshape(o, kt * (c + cp))
            wh = wm.half().float(); xh = cols.half().float()
            y = F.linear(xh, wh) + (F.linear(mx_quant(xh, ef), mx_quant(wm - wh, ef)) + F.linear(mx_quant(cols - xh, ef), mx_quant(wh, ef)))
            return (y + b).permute(0, 2, 1)
        if fmt.endswith("x2w") or fmt.endswith("x2a"):
            base = torch.bfloat16 if fmt.startswith("bf16") else torch.float16
            wh = w.to(base).float(); xh = x.to(base).float()
            if fmt.endswith("x2w"):
                return F.conv1d(xh, wh, b, **kw) + F.conv1d(xh, (w - wh).to(base).float(), None, **kw)
            return F.conv1d(xh, wh, b, **kw) + F.conv1d((x - xh).to(base).float(), wh, None, **kw)
        if fmt.endswith("x3"):
            base = torch.bfloat16 if fmt.startswith("bf16") else torch.float16
            wh = w.to(base).float(); wl = (w - wh).to(base).float()
            xh = x.to(base).float(); xl = (x - xh).to(base).float()
            return F.conv1d(xh, wh, b, **kw) + (F.conv1d(xh, wl, None, **kw) + F.conv1d(xl, wh, None, **kw))
        return F.conv1d(rnd(x, fmt), rnd(w, fmt), b, **kw)


def attn(q, k, v, pfmt, mask=None):
    w = torch.matmul(q, k.transpose(-1, -2))
    if mask is not None:
        w = w + mask
    w = rnd(torch.softmax(w, dim=-1), pfmt)
    o = torch.matmul(w, v)
    b, h, t, e = o.shape
    return o.transpose(1, 2).reshape(b, t, h * e)


@torch.no_grad()
def encoder_forward(P, sd, cfg, feats):
    p = "model.encoder."
    x = F.gelu(P.conv(feats, sd[p + "conv1.weight"], sd[p + "conv1.bias"], padding=1))
    x = F.gelu(P.conv(x, sd[p + "conv2.weight"], sd[p + "conv2.bias"], stride=2, padding=1))
    x = x.permute(0, 2, 1) + sd[p + "embed_positions.weight"][None, : x.shape[-1]]
    scale = (cfg.d_model // cfg.heads) ** -0.5
    ef = P.fmt["eattn"]
    for i in range(cfg.encoder_layers):
        lp = f"{p}layers.{i}."
        r = x
        y = W._ln(x, sd, lp + "self_attn_layer_norm")
        q = rnd(W._heads(P.linear(y, sd, lp + "self_attn.q_proj") * scale, cfg.heads), ef)
        k = rnd(W._heads(P.linear(y, sd, lp + "self_attn.k_proj", bias=False), cfg.heads), ef)
        v = rnd(W._heads(P.linear(y, sd, lp + "self_attn.v_proj"), cfg.heads), ef)
        x = r + P.linear(attn(q, k, v, ef), sd, lp + "self_attn.out_proj")
        r = x
        y = W._ln(x, sd, lp + "final_layer_norm")
        y = F.gelu(P.linear(y, sd, lp + "fc1"))
        x = r + P.linear(y, sd, lp + "fc2")
    return W._ln(x, sd, p + "layer_norm")


class Decoder(W.Decoder):
    def __init__(self, P, sd, cfg, enc_out):
        self.P = P
        self.sd, self.cfg = sd, cfg
        self.p = "model.decoder."
        self.scale = (cfg.d_model // cfg.heads) ** -0.5
        self.cross = []
        for i in range(cfg.decoder_layers):
            lp = f"{self.p}layers.{i}.encoder_attn."
            k = rnd(W._heads(P.linear(enc_out, sd, lp + "k_proj", bias=False), cfg.heads), P.fmt["ck"])
            v = rnd(W._heads(P.linear(enc_out, sd, lp + "v_proj"), cfg.heads), P.fmt["cv"])
            self.cross.append((k, v))
        self.self_kv = [None] * cfg.decoder_layers
        self.pos = 0
        self.margins = None

    @torch.no_grad()
    def step(self, tokens):
        sd, cfg, p, P = self.sd, self.cfg, self.p, self.P
        n = tokens.shape[1]
        x = sd[p + "embed_tokens.weight"][tokens] + sd[p + "embed_positions.weight"][self.pos:self.pos + n][None]
        mask = None
        if n > 1:
            mask = torch.full((n, self.pos + n), float("-inf"))
            mask = torch.triu(mask, diagonal=self.pos + 1)
        for i in range(cfg.decoder_layers):
            lp = f"{p}layers.{i}."
            r = x
            y = W._ln(x, sd, lp + "self_attn_layer_norm")
            q = rnd(W._heads(P.linear(y, sd, lp + "self_attn.q_proj") * self.scale, cfg.heads), P.fmt["dq"])
            k = rnd(W._heads(P.linear(y, sd, lp + "self_attn.k_proj", bias=False), cfg.heads), P.fmt["skv"])
            v = rnd(W._heads(P.linear(y, sd, lp + "self_attn.v_proj"), cfg.heads), P.fmt["skv"])
            if self.self_kv[i] is not None:
                k = torch.cat([self.self_kv[i][0], k], dim=2)
                v = torch.cat([self.self_kv[i][1], v], dim=2)
            self.self_kv[i] = (k, v)
            x = r + P.linear(attn(q, k, v, "f32", mask), sd, lp + "self_attn.out_proj")
            r = x
            y = W._ln(x, sd, lp + "encoder_attn_layer_norm")
            q = rnd(W._heads(P.linear(y, sd, lp + "encoder_attn.q_proj") * self.scale, cfg.heads), P.fmt["dq"])
            ck, cv = self.cross[i]
            x = r + P.linear(attn(q, ck, cv, "f32"), sd, lp + "encoder_attn.out_proj")
            r = x
            y = W._ln(x, sd, lp + "final_layer_norm")
            y = F.gelu(P.linear(y, sd, lp + "fc1"))
            x = r + P.linear(y, sd, lp + "fc2")
        self.pos += n
        x = W._ln(x[:, -1], sd, p + "layer_norm")
        logits = P.linear(x, sd, p + "embed_tokens", bias=False).float()
        if self.margins is not None:
            self.margins.append(logits.clone())
        return logits


MARGINS = []      # (margin, top1 is a time token, top1 id) of every decode row-step of the fp32 oracle (policy "margins")


class OracleSegmenter(SegmenterBase):
    """SegmenterBase with the device stages replaced by the CPU oracle under a precision policy."""

    def __init__(self, policy, collect_margins=False, model_dir=None):
        super().__init__()
        from safetensors.torch import load_file
        self.P = policy
        model_dir = model_dir or MODEL_DIR
        self.sd = {k: v.float() for k, v in load_file(os.path.join(model_dir, "model.safetensors")).items()}
        with open(os.path.join(model_dir, "config.json")) as f:
            hf = json.load(f)
        self.cfg = W.RefConfig.from_hf_dict(hf)
        self._adopt_config(hf)
        self.tokenizer = WhisperSegTokenizer.from_pretrained(model_dir, language="english")
        self.device_list = ["cpu-oracle"]
        self.collect = collect_margins
        self.feat_cache = {}

    def get_sliced_audios_features(self, audio, sr, min_frequency, spec_time_step, num_trials):
        key = (hash(audio.tobytes()), sr, min_frequency, spec_time_step, num_trials)
        if key not in self.feat_cache:
            self.feat_cache[key] = FE.sliced_audio_features(audio, sr, min_frequency, spec_time_step, num_trials)
        return self.feat_cache[key]

    def generate_segment_text(self, sliced, batch_size, max_length, num_beams, top_k=1, top_p=1.0, length_penalty=1.0,
                              status_monitor=None):
        feats = torch.from_numpy(np.stack([s[2] for s in sliced]))
        gp = W.GenParams(prompt=TM.PROMPT, eos_token_id=TM.EOT, pad_token_id=TM.EOT, max_length=max_length, num_beams=num_beams,
                         length_penalty=length_penalty, suppress_tokens=TM.SUPPRESS, begin_suppress_tokens=TM.BEGIN_SUPPRESS)
        P = self.P
        texts = []
        # window by window: a window's tokens must not depend on its batch (HF stops a batch when all items are done; the
        # per-item result is frozen before that, so single-window decoding equals batched decoding)
        for i in range(feats.shape[0]):
            orig_enc, orig_dec = W.encoder_forward, W.Decoder
            holder = {}

            def make_dec(sd, cfg, enc_out, P=P, holder=holder):
                d = Decoder(P, sd, cfg, enc_out)
                if self.collect:
                    d.margins = []
                holder["d"] = d
                return d
            W.encoder_forward = lambda sd, cfg, f, P=P: encoder_forward(P, sd, cfg, f)
            W.Decoder = make_dec
            try:
                out = W.generate(self.sd, self.cfg, feats[i:i + 1], gp)
            finally:
                W.encoder_forward, W.Decoder = orig_enc, orig_dec
            if self.collect:
                for step, lg in enumerate(holder["d"].margins):
                    lg = lg.clone()
                    lg[:, TM.SUPPRESS] = float("-inf")
                    if step == 0:
                        lg[:, TM.BEGIN_SUPPRESS] = float("-inf")
                    top = torch.topk(lg, 2, dim=-1)
                    for r in range(lg.shape[0]):
                        t1 = int(top.indices[r, 0])
                        MARGINS.append((float(top.values[r, 0] - top.values[r, 1]), TM.TIME0 <= t1 <= TM.TIME0 + 1000, t1))
            row = out[0].tolist()
            toks = list(TM.PROMPT) + W.canonical(row, 3, TM.EOT, TM.PROMPT)
            texts.append(self.tokenizer.batch_decode([toks], skip_special_tokens=False)[0])
        return texts


def main():
    from tools.parity_sweep import score
    torch.set_num_threads(int(os.environ.get("STUDY_THREADS", 8)))
    dest = sys.argv[1]
    with open(os.path.join(ROOT, "tests", "golden", SWEEP_ROWS)) as f:
        sweep = json.load(f)
    limit = int(os.environ.get("STUDY_RUNS", len(sweep)))
    sweep = sweep[:limit]
    res = {}
    if os.path.exists(dest):
        with open(dest) as f:
            res = json.load(f)
    for text in sys.argv[2:]:
        if text.startswith("logiterr:"):
            # first-step logits of 8 sweep recordings' first windows under the policy against the fp32 oracle
            pol = text[len("logiterr:"):]
            sego, segp = OracleSegmenter(Policy("")), OracleSegmenter(Policy(pol))
            errs, scale = [], 0.0
            for run in sweep[:32:4]:
                audio = GI.tiny_recording(run["seed"], run["n_windows"], variant=SWEEP_VARIANT)
                sl = sego.get_sliced_audios_features(audio, TM.SR, 0, TM.STS, 1)
                feats = torch.from_numpy(np.stack([s_[2] for s_ in sl[:2]]))
                outs = []
                for P_ in (sego.P, segp.P):
                    enc = encoder_forward(P_, sego.sd, sego.cfg, feats)
                    dec = Decoder(P_, sego.sd, sego.cfg, enc)
                    outs.append(dec.step(torch.tensor([TM.PROMPT] * feats.shape[0])))
                errs.append(float((outs[0] - outs[1]).abs().max()))
                scale = max(scale, float(outs[0].abs().max()))
            res[text] = {"max_abs_err": max(errs), "mean_max_abs_err": float(np.mean(errs)), "logit_scale": scale}
            print(text, res[text], flush=True)
            with open(dest, "w") as f:
                json.dump(res, f, indent=1)
            continue
        if text == "margins":
            seg = OracleSegmenter(Policy(""), collect_margins=True)
            r = score(seg, sweep, SWEEP_VARIANT)
            m = np.array([x[0] for x in MARGINS])
            is_time = np.array([x[1] for x in MARGINS])
            edges = [0, 1e-5, 1e-4, 3e-4, 1e-3, 3e-3, 1e-2, 3e-2, 0.1, 0.3, 1.0, 3.0, 1e9]
            res["margins"] = dict(
                runs=r["runs"], exact_runs=r["exact_runs"], row_steps=int(len(m)), time_token_row_steps=int(is_time.sum()),
                edges=edges, hist_all=np.histogram(m, edges)[0].tolist(), hist_time_tokens=np.histogram(m[is_time], edges)[0].tolist(),
                quantiles_time_tokens={str(q): float(np.quantile(m[is_time], q)) for q in (0.001, 0.01, 0.05, 0.25, 0.5)})
            print("margins", json.dumps(res["margins"]), flush=True)
        else:
            seg = OracleSegmenter(Policy(text))
            r = score(seg, sweep, SWEEP_VARIANT)
            res[text] = r
            print(text, "runs", r["runs"], "exact", r["exact_runs"], "within +-1 frame", r["within_tolerance_runs"],
                  "structure mismatches", len(r["structure_mismatch_runs"]), "hist", r["frame_hist"], flush=True)
        with open(dest, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()

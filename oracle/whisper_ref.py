"""ORACLE (test infrastructure, NOT product code) — CPU restatement of the Whisper
encoder-decoder forward and of HuggingFace's greedy / beam-search decoding, in plain
torch fp32 on the host.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

The reference's arithmetic for this part of the path lives in a THIRD-PARTY dependency
that is not under /root/reference: `transformers` (pinned 4.38.2 in the reference's
requirements.txt:1; 5.15.0 is what is installed in the image).  Call sites in the
reference: model.py:609-619 and model.py:655-666 (`model.generate(...)`).  What is
restated here, citing the installed HF sources:

  encoder_forward     HF models/whisper/modeling_whisper.py:592-646 (encoder), :360-413 (layer),
                      :241-357 (attention; q scaled by head_dim**-0.5, k_proj has no bias)
  Decoder.step        HF modeling_whisper.py:690-796 (decoder), :416-506 (layer), :970,1080 (tied proj_out)
  generate (beams>1)  HF generation/utils.py:3208-3510 (_beam_search) + helpers :3008-3206
  generate (beams==1) HF generation/utils.py `_sample` with do_sample False (the reference passes
                      do_sample=True, top_k=1 which is the deterministic argmax, model.py:615-616)
  logits processors   HF generation/logits_process.py: SuppressTokensAtBeginLogitsProcessor (:1816),
                      SuppressTokensLogitsProcessor (:1869), wired by generation_whisper.py:1774-1812

Pinning: tests/test_oracle_model.py checks this file against tests/golden/tiny_*.npz —
encoder outputs, first-step logits and token sequences captured from HF `generate` driven
through the reference's own WhisperSegmenterForEval (tools/make_golden.py).
"""
import math
from dataclasses import dataclass, field

import torch
import torch.nn.functional as F


@dataclass
class RefConfig:
    d_model: int
    encoder_layers: int
    decoder_layers: int
    heads: int
    ffn: int
    vocab_size: int
    n_mels: int = 80
    max_source_positions: int = 500
    max_target_positions: int = 448

    @staticmethod
    def from_hf_dict(c):
        return RefConfig(
            d_model=c["d_model"], encoder_layers=c["encoder_layers"], decoder_layers=c["decoder_layers"],
            heads=c["encoder_attention_heads"], ffn=c["encoder_ffn_dim"], vocab_size=c["vocab_size"],
            n_mels=c.get("num_mel_bins", 80), max_source_positions=c.get("max_source_positions", 500),
            max_target_positions=c.get("max_target_positions", 448))


@dataclass
class GenParams:
    prompt: list
    eos_token_id: int
    pad_token_id: int
    max_length: int = 448
    num_beams: int = 4
    length_penalty: float = 1.0
    suppress_tokens: list = field(default_factory=list)
    begin_suppress_tokens: list = field(default_factory=list)


def _lin(x, sd, prefix, bias=True):
    return F.linear(x, sd[prefix + ".weight"], sd[prefix + ".bias"] if bias else None)


def _ln(x, sd, prefix):
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"], 1e-5)


def _heads(x, h):
    b, t, d = x.shape
    return x.view(b, t, h, d // h).transpose(1, 2)


def _attn(q, k, v, mask=None):
    """q already scaled.  q [B,H,Tq,64], k/v [B,H,Tk,64]."""
    w = torch.matmul(q, k.transpose(-1, -2))
    if mask is not None:
        w = w + mask
    w = torch.softmax(w, dim=-1)
    o = torch.matmul(w, v)
    b, h, t, e = o.shape
    return o.transpose(1, 2).reshape(b, t, h * e)


@torch.no_grad()
def encoder_forward(sd, cfg, feats, return_hidden=False):
    """feats float32 [B,80,1000] -> [B,500,d]."""
    p = "model.encoder."
    x = F.gelu(F.conv1d(feats, sd[p + "conv1.weight"], sd[p + "conv1.bias"], padding=1))
    x = F.gelu(F.conv1d(x, sd[p + "conv2.weight"], sd[p + "conv2.bias"], stride=2, padding=1))
    x = x.permute(0, 2, 1) + sd[p + "embed_positions.weight"][None, : x.shape[-1]]
    scale = (cfg.d_model // cfg.heads) ** -0.5
    hidden = [x]
    for i in range(cfg.encoder_layers):
        lp = f"{p}layers.{i}."
        r = x
        y = _ln(x, sd, lp + "self_attn_layer_norm")
        q = _heads(_lin(y, sd, lp + "self_attn.q_proj") * scale, cfg.heads)
        k = _heads(_lin(y, sd, lp + "self_attn.k_proj", bias=False), cfg.heads)
        v = _heads(_lin(y, sd, lp + "self_attn.v_proj"), cfg.heads)
        x = r + _lin(_attn(q, k, v), sd, lp + "self_attn.out_proj")
        r = x
        y = _ln(x, sd, lp + "final_layer_norm")
        y = F.gelu(_lin(y, sd, lp + "fc1"))
        x = r + _lin(y, sd, lp + "fc2")
        hidden.append(x)
    x = _ln(x, sd, p + "layer_norm")
    return (x, hidden) if return_hidden else x


class Decoder:
    """Incremental decoder with self-attention KV cache; cross K/V computed once per row."""

    def __init__(self, sd, cfg, enc_out):
        self.sd, self.cfg = sd, cfg
        self.p = "model.decoder."
        self.scale = (cfg.d_model // cfg.heads) ** -0.5
        self.cross = []
        for i in range(cfg.decoder_layers):
            lp = f"{self.p}layers.{i}.encoder_attn."
            k = _heads(_lin(enc_out, sd, lp + "k_proj", bias=False), cfg.heads)
            v = _heads(_lin(enc_out, sd, lp + "v_proj"), cfg.heads)
            self.cross.append((k, v))
        self.self_kv = [None] * cfg.decoder_layers
        self.pos = 0

    def reorder(self, idx):
        self.self_kv = [(k.index_select(0, idx), v.index_select(0, idx)) for (k, v) in self.self_kv]

    @torch.no_grad()
    def step(self, tokens):
        """tokens int64 [R, n] (n = prompt length on the first call, then 1) -> logits fp32 [R, V] of the last position."""
        sd, cfg, p = self.sd, self.cfg, self.p
        n = tokens.shape[1]
        x = sd[p + "embed_tokens.weight"][tokens] + sd[p + "embed_positions.weight"][self.pos:self.pos + n][None]
        mask = None
        if n > 1:
            mask = torch.full((n, self.pos + n), float("-inf"))
            mask = torch.triu(mask, diagonal=self.pos + 1)
        for i in range(cfg.decoder_layers):
            lp = f"{p}layers.{i}."
            r = x
            y = _ln(x, sd, lp + "self_attn_layer_norm")
            q = _heads(_lin(y, sd, lp + "self_attn.q_proj") * self.scale, cfg.heads)
            k = _heads(_lin(y, sd, lp + "self_attn.k_proj", bias=False), cfg.heads)
            v = _heads(_lin(y, sd, lp + "self_attn.v_proj"), cfg.heads)
            if self.self_kv[i] is not None:
                k = torch.cat([self.self_kv[i][0], k], dim=2)
                v = torch.cat([self.self_kv[i][1], v], dim=2)
            self.self_kv[i] = (k, v)
            x = r + _lin(_attn(q, k, v, mask), sd, lp + "self_attn.out_proj")
            r = x
            y = _ln(x, sd, lp + "encoder_attn_layer_norm")
            q = _heads(_lin(y, sd, lp + "encoder_attn.q_proj") * self.scale, cfg.heads)
            ck, cv = self.cross[i]
            x = r + _lin(_attn(q, ck, cv), sd, lp + "encoder_attn.out_proj")
            r = x
            y = _ln(x, sd, lp + "final_layer_norm")
            y = F.gelu(_lin(y, sd, lp + "fc1"))
            x = r + _lin(y, sd, lp + "fc2")
        self.pos += n
        x = _ln(x[:, -1], sd, p + "layer_norm")
        return F.linear(x, sd[p + "embed_tokens.weight"]).float()


def _process(scores, cur_len, gp):
    """SuppressTokens (every step) + SuppressTokensAtBegin (only when cur_len == len(prompt))."""
    if gp.suppress_tokens:
        scores[:, gp.suppress_tokens] = float("-inf")
    if gp.begin_suppress_tokens and cur_len == len(gp.prompt):
        scores[:, gp.begin_suppress_tokens] = float("-inf")
    return scores


def _gather(t, idx):
    while idx.dim() < t.dim():
        idx = idx.unsqueeze(-1)
    return torch.take_along_dim(t, idx, dim=1)


@torch.no_grad()
def generate(sd, cfg, feats, gp, return_first_logits=False):
    """feats [B,80,1000] -> int64 [B, L] token ids INCLUDING the prompt, padded with pad_token_id.

    Follows HF generate as the reference calls it (model.py:609-619).
    """
    B = feats.shape[0]
    enc = encoder_forward(sd, cfg, feats)
    P = len(gp.prompt)
    prompt = torch.tensor(gp.prompt, dtype=torch.int64)
    first_logits = None

    if gp.num_beams == 1:
        dec = Decoder(sd, cfg, enc)
        seq = prompt[None].repeat(B, 1)
        unfinished = torch.ones(B, dtype=torch.int64)
        cur_len = P
        inp = seq
        while True:
            logits = dec.step(inp)
            if first_logits is None:
                first_logits = logits.clone()
            scores = _process(logits.clone(), cur_len, gp)
            nxt = torch.argmax(scores, dim=-1)
            nxt = nxt * unfinished + gp.pad_token_id * (1 - unfinished)
            seq = torch.cat([seq, nxt[:, None]], dim=1)
            cur_len += 1
            unfinished = unfinished & (nxt != gp.eos_token_id).long()
            if cur_len >= gp.max_length:
                unfinished = unfinished * 0
            if unfinished.max() == 0:
                break
            inp = nxt[:, None]
        return (seq, first_logits) if return_first_logits else seq

    nb = gp.num_beams
    V = cfg.vocab_size
    K = 2 * nb  # one EOS id -> beams_to_keep = max(2, 1 + 1) * num_beams
    max_length = gp.max_length
    lp = gp.length_penalty
    dec = Decoder(sd, cfg, enc.repeat_interleave(nb, dim=0))
    running_sequences = torch.full((B, nb, max_length), gp.pad_token_id, dtype=torch.int64)
    running_sequences[:, :, :P] = prompt
    sequences = running_sequences.clone()
    running_beam_scores = torch.zeros((B, nb), dtype=torch.float32)
    running_beam_scores[:, 1:] = -1e9
    beam_scores = torch.full((B, nb), -1e9, dtype=torch.float32)
    is_sent_finished = torch.zeros((B, nb), dtype=torch.bool)
    unsat = torch.ones((B, 1), dtype=torch.bool)  # is_early_stop_heuristic_unsatisfied
    top_mask = torch.cat([torch.ones(nb, dtype=torch.bool), torch.zeros(K - nb, dtype=torch.bool)])
    gen_len = torch.zeros((B, nb), dtype=torch.int64)          # generated length of each finished slot
    cur_len = P
    inp = running_sequences[:, :, :P].reshape(B * nb, P)
    while True:
        logits = dec.step(inp)
        if first_logits is None:
            first_logits = logits.clone()
        log_probs = F.log_softmax(logits, dim=-1)
        log_probs = _process(log_probs, cur_len, gp)
        log_probs = log_probs.view(B, nb, V) + running_beam_scores[:, :, None]
        log_probs = log_probs.reshape(B, nb * V)
        topk_log_probs, topk_indices = torch.topk(log_probs, k=K)
        topk_beam = topk_indices // V
        topk_ids = topk_indices % V
        topk_running_sequences = _gather(running_sequences, topk_beam)
        topk_running_sequences[:, :, cur_len] = topk_ids
        hits = (topk_ids == gp.eos_token_id) | (cur_len + 1 >= max_length)
        # e. running beams for the next iteration
        topk_running_log_probs = topk_log_probs + hits.to(torch.float32) * -1.0e9
        next_idx = torch.topk(topk_running_log_probs, k=nb)[1]
        running_sequences = _gather(topk_running_sequences, next_idx)
        running_beam_scores = _gather(topk_running_log_probs, next_idx)
        beam_src = _gather(topk_beam, next_idx)
        # f. finished beams
        did_finish = hits & top_mask[None, :]
        fin_scores = topk_log_probs / ((cur_len + 1 - P) ** lp)
        fin_scores = fin_scores + (~unsat).to(torch.float32) * -1.0e9
        fin_scores = fin_scores + (~did_finish) * -1.0e9
        merged_sequences = torch.cat((sequences, topk_running_sequences), dim=1)
        merged_scores = torch.cat((beam_scores, fin_scores), dim=1)
        merged_fin = torch.cat((is_sent_finished, did_finish), dim=1)
        merged_len = torch.cat((gen_len, torch.full((B, K), cur_len + 1 - P, dtype=torch.int64)), dim=1)
        mi = torch.topk(merged_scores, k=nb)[1]
        sequences = _gather(merged_sequences, mi)
        beam_scores = _gather(merged_scores, mi)
        is_sent_finished = _gather(merged_fin, mi)
        gen_len = _gather(merged_len, mi)
        # g. cache reorder + stopping condition
        flat_src = (beam_src + torch.arange(B)[:, None] * nb).reshape(-1)
        dec.reorder(flat_src)
        cur_len += 1
        best_running = running_beam_scores[:, :1] / ((cur_len - P) ** lp)
        worst_finished = torch.where(is_sent_finished, torch.min(beam_scores, dim=1, keepdim=True)[0],
                                     torch.tensor(-1.0e9))
        unsat = unsat & torch.any(best_running > worst_finished, dim=-1, keepdim=True)
        if not (bool(torch.any(unsat)) and not bool(torch.all(hits))):
            break
        inp = running_sequences[:, :, cur_len - 1].reshape(B * nb, 1)
    out_len = int((gen_len[:, 0]).max()) + P
    out = sequences[:, 0, :out_len]
    return (out, first_logits) if return_first_logits else out


@torch.no_grad()
def score_sequence(sd, cfg, feats, gp, tokens):
    """HF's beam score of ONE given hypothesis for ONE window: sum of log_softmax(logits)[token] over the generated tokens
    (teacher-forced through the same decoder) / generated_length ** length_penalty — the quantity `_beam_search` ranks finished
    hypotheses by (HF generation/utils.py:3440-3452: topk_log_probs / (cur_len + 1 - prompt_len) ** length_penalty).
    feats [1,80,1000]; tokens: ids INCLUDING the prompt, up to and including EOS if there is one (pads stripped).
    Used by tests to show that a beam result that differs from the oracle's is an equally scored hypothesis (a near-tie
    resolved differently by another summation order), not an error."""
    P = len(gp.prompt)
    toks = [int(t) for t in tokens]
    assert toks[:P] == list(gp.prompt)
    gen = toks[P:]
    if gp.eos_token_id in gen:
        gen = gen[: gen.index(gp.eos_token_id) + 1]
    dec = Decoder(sd, cfg, encoder_forward(sd, cfg, feats))
    total = 0.0
    inp = torch.tensor([toks[:P]], dtype=torch.int64)
    for t in gen:
        lp = F.log_softmax(dec.step(inp), dim=-1)
        total += float(lp[0, t])
        inp = torch.tensor([[t]], dtype=torch.int64)
    return total / (len(gen) ** gp.length_penalty)


def canonical(tokens, prompt_len, eos_token_id, prompt=None):
    """Generated ids up to and including the first EOS, prompt stripped (works for both HF
    conventions: 4.38.2 returns the prompt, 5.15 strips it)."""
    toks = [int(t) for t in tokens]
    if prompt is not None and toks[:len(prompt)] == list(prompt):
        toks = toks[len(prompt):]
    elif prompt is None:
        toks = toks[prompt_len:]
    out = []
    for t in toks:
        out.append(t)
        if t == eos_token_id:
            break
    return out


def random_state_dict(cfg, seed=0, std=0.02, dtype=torch.float32, fast=False):
    """Seeded random weights with HF parameter names (no checkpoints exist offline).

    fast=True tiles one 8M-sample normal pool instead of drawing every element (about 40x quicker for the
    1.5 B-parameter geometry); used only where the values do not matter (bench.py's cpu_baseline timing)."""
    g = torch.Generator().manual_seed(seed)
    d, f = cfg.d_model, cfg.ffn
    pool = torch.randn(1 << 23, generator=g) if fast else None
    pool2 = torch.cat([pool, pool]) if fast else None
    cursor = [0]

    def rn(*shape, s=std):
        if pool is None:
            return (torch.randn(*shape, generator=g) * s).to(dtype)
        n = 1
        for v in shape:
            n *= v
        reps = (n + pool.numel() - 1) // pool.numel() + 1
        start = cursor[0] % pool.numel()
        cursor[0] += 7919 + n
        flat = pool.repeat(reps)[start:start + n] if reps > 2 else pool2[start:start + n]
        return (flat.reshape(*shape) * s).to(dtype)

    sd = {}
    e = "model.encoder."
    sd[e + "conv1.weight"] = rn(d, cfg.n_mels, 3, s=0.05)
    sd[e + "conv1.bias"] = rn(d)
    sd[e + "conv2.weight"] = rn(d, d, 3)
    sd[e + "conv2.bias"] = rn(d)
    sd[e + "embed_positions.weight"] = rn(cfg.max_source_positions, d)

    def attn(prefix):
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            sd[f"{prefix}{n}.weight"] = rn(d, d, s=d ** -0.5)
            if n != "k_proj":
                sd[f"{prefix}{n}.bias"] = rn(d)

    def ln(prefix):
        sd[prefix + ".weight"] = (1.0 + rn(d, s=0.1)).to(dtype)
        sd[prefix + ".bias"] = rn(d, s=0.1)

    def mlp(prefix):
        sd[prefix + "fc1.weight"] = rn(f, d, s=d ** -0.5)
        sd[prefix + "fc1.bias"] = rn(f)
        sd[prefix + "fc2.weight"] = rn(d, f, s=f ** -0.5)
        sd[prefix + "fc2.bias"] = rn(d)

    for i in range(cfg.encoder_layers):
        lp = f"{e}layers.{i}."
        attn(lp + "self_attn.")
        ln(lp + "self_attn_layer_norm")
        mlp(lp)
        ln(lp + "final_layer_norm")
    ln(e + "layer_norm")
    dd = "model.decoder."
    sd[dd + "embed_tokens.weight"] = rn(cfg.vocab_size, d, s=0.05)
    sd[dd + "embed_positions.weight"] = rn(cfg.max_target_positions, d, s=0.02)
    for i in range(cfg.decoder_layers):
        lp = f"{dd}layers.{i}."
        attn(lp + "self_attn.")
        ln(lp + "self_attn_layer_norm")
        attn(lp + "encoder_attn.")
        ln(lp + "encoder_attn_layer_norm")
        mlp(lp)
        ln(lp + "final_layer_norm")
    ln(dd + "layer_norm")
    return sd

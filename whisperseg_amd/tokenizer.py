"""Minimal id <-> text mapping for WhisperSeg checkpoints (replaces HF WhisperTokenizer on the path:
reference model.py:637, 656, 667).  Only what segmentation needs: token -> id for the 3-token prompt,
pad/eos ids, and `batch_decode(ids, skip_special_tokens=False)` producing the text the segment regex
scans (added tokens verbatim, byte-level BPE tokens through the GPT-2 byte decoder, nothing inserted
between tokens).

Reads `vocab.json` + `added_tokens.json` (slow-tokenizer layout, as saved by the reference's
tokenizer.save_pretrained, model.py:66) or `tokenizer.json` (fast layout).
"""
import json
import os


def _bytes_to_unicode():
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    return {chr(c): b for b, c in zip(bs, cs)}


class WhisperSegTokenizer:
    def __init__(self, vocab, added_tokens, eos_token="<|endoftext|>", pad_token=None):
        self.encoder = dict(vocab)
        self.encoder.update(added_tokens)
        self.added = set(added_tokens)
        self.decoder = {i: t for t, i in self.encoder.items()}
        self.byte_decoder = _bytes_to_unicode()
        if eos_token not in self.encoder:
            raise ValueError(f"{eos_token} not in the vocabulary")
        self.eos_token_id = self.encoder[eos_token]
        self.pad_token_id = self.encoder[pad_token] if pad_token else self.eos_token_id

    @classmethod
    def from_pretrained(cls, model_dir, language="english"):
        vocab, added = {}, {}
        vj, aj, tj = (os.path.join(model_dir, n) for n in ("vocab.json", "added_tokens.json", "tokenizer.json"))
        if os.path.exists(vj):
            with open(vj, encoding="utf-8") as f:
                vocab = json.load(f)
            if os.path.exists(aj):
                with open(aj, encoding="utf-8") as f:
                    added = json.load(f)
        elif os.path.exists(tj):
            with open(tj, encoding="utf-8") as f:
                t = json.load(f)
            vocab = t["model"]["vocab"]
            added = {a["content"]: a["id"] for a in t.get("added_tokens", [])}
        else:
            raise FileNotFoundError(f"no vocab.json / tokenizer.json under {model_dir}")
        added = {k: v for k, v in added.items()}
        for k in list(vocab):
            if k.startswith("<|") and k.endswith("|>"):
                added.setdefault(k, vocab[k])
        pad = None
        sm = os.path.join(model_dir, "special_tokens_map.json")
        if os.path.exists(sm):
            with open(sm, encoding="utf-8") as f:
                m = json.load(f)
            p = m.get("pad_token")
            pad = p["content"] if isinstance(p, dict) else p
        if pad is not None and pad not in vocab and pad not in added:
            pad = None
        return cls(vocab, added, pad_token=pad)

    def convert_tokens_to_ids(self, tokens):
        if isinstance(tokens, str):
            return self.encoder[tokens]
        return [self.encoder[t] for t in tokens]

    def decode(self, ids, skip_special_tokens=False):
        out, buf = [], bytearray()

        def flush():
            if buf:
                out.append(buf.decode("utf-8", errors="replace"))
                buf.clear()

        for i in ids:
            tok = self.decoder.get(int(i))
            if tok is None:
                continue
            if tok in self.added:
                if skip_special_tokens:      # HF drops them BEFORE byte decoding: the bytes either side form one UTF-8 run
                    continue
                flush()
                out.append(tok)
            else:
                buf.extend(self.byte_decoder.get(ch, ord("?")) for ch in tok)
        flush()
        return "".join(out)

    def batch_decode(self, batch_ids, skip_special_tokens=False):
        return [self.decode(ids, skip_special_tokens) for ids in batch_ids]

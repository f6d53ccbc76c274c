"""Pins whisperseg_amd.tokenizer on HuggingFace's WhisperTokenizer.

Builds a `transformers.WhisperTokenizer` (5.15, the wheel of this image) from a small synthetic byte-level BPE vocabulary whose
digits merge into multi-digit tokens as in the real Whisper vocabulary, adds `<|0|>`..`<|1000|>` and the species tokens exactly
as the reference does (model.py:111-113: add_tokens(..., special_tokens=True)), saves it the way `save_pretrained` writes it
(reference model.py:66) and records `batch_decode(ids, skip_special_tokens=False / True)` for random id rows.

    python tools/make_tok_fixture.py            -> tests/golden/tok_fixture/{hf/, slow/, decode_cases.json}

hf/    tokenizer.json + tokenizer_config.json — what transformers 5.15 writes.
slow/  vocab.json + merges.txt + added_tokens.json + special_tokens_map.json — the slow-tokenizer layout transformers 4.38.2
       (the reference's pin, requirements.txt:1) writes; 5.15 no longer emits it, so it is derived here from the same
       tokenizer object (BPE vocabulary without the added tokens / the added tokens with their ids).
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import tiny_model as TM  # noqa: E402

OUT = os.environ.get("WSEG_TOK_FIXTURE_OUT") or os.path.join(ROOT, "tests", "golden", "tok_fixture")
SPECIALS = ["<|endoftext|>", "<|startoftranscript|>", "<|en|>", "<|de|>", "<|translate|>", "<|transcribe|>", "<|startoflm|>",
            "<|startofprev|>", "<|nospeech|>", "<|notimestamps|>"]
SPECIES = ["<|zebra_finch|>", "<|bengalese_finch|>", "<|mouse|>", "<|marmoset|>", "<|human|>", "<|unknown|>", "<|animal|>"]


def build():
    from transformers import WhisperTokenizer
    vocab = dict(TM.base_vocab())          # the 256 byte tokens in GPT-2 order (digits '0'..'9' = ids 15..24)
    merges = []

    def add(a, b):
        merges.append((a, b))
        vocab.setdefault(a + b, len(vocab))
    for a in "0123456789":                 # every two-digit token, some three-digit ones, a few words with the space marker
        for b in "0123456789":
            add(a, b)
    for ab in ("10", "12", "25", "99", "00"):
        for c in "0579":
            add(ab, c)
    for a, b in (("Ġ", "a"), ("t", "h"), ("th", "e"), ("Ġ", "the"), ("i", "n"), ("Ġ", "1"), ("Ġ1", "2")):
        add(a, b)
    n_bpe = len(vocab)
    for sp in SPECIALS:
        vocab[sp] = len(vocab)
    tok = WhisperTokenizer(vocab=vocab, merges=merges, language="english", additional_special_tokens=SPECIALS[1:])
    tok.add_tokens(["<|%d|>" % i for i in range(1001)], special_tokens=True)          # reference model.py:112
    tok.add_tokens(SPECIES, special_tokens=True)                                       # reference model.py:113
    return tok, vocab, merges, n_bpe


def main():
    tok, vocab, merges, n_bpe = build()
    os.makedirs(os.path.join(OUT, "hf"), exist_ok=True)
    os.makedirs(os.path.join(OUT, "slow"), exist_ok=True)
    tok.save_pretrained(os.path.join(OUT, "hf"))
    # get_vocab() of a fast tokenizer iterates a Rust hash map (a different order every process): everything derived from it is
    # ordered by token id, so that two runs of this script write the same bytes (tests/test_tokenizer_wav.py asserts it)
    full = dict(sorted(tok.get_vocab().items(), key=lambda kv: kv[1]))
    bpe = {t: i for t, i in full.items() if i < n_bpe}
    added = {t: i for t, i in full.items() if i >= n_bpe}
    with open(os.path.join(OUT, "slow", "vocab.json"), "w", encoding="utf-8") as f:
        json.dump(bpe, f, ensure_ascii=False)
    with open(os.path.join(OUT, "slow", "merges.txt"), "w", encoding="utf-8") as f:
        f.write("#version: 0.2\n" + "\n".join(a + " " + b for a, b in merges) + "\n")
    with open(os.path.join(OUT, "slow", "added_tokens.json"), "w", encoding="utf-8") as f:
        json.dump(added, f, ensure_ascii=False)
    with open(os.path.join(OUT, "slow", "special_tokens_map.json"), "w", encoding="utf-8") as f:
        json.dump({"bos_token": "<|endoftext|>", "eos_token": "<|endoftext|>", "unk_token": "<|endoftext|>", "pad_token": "<|endoftext|>"}, f)
    # id rows: the reference's label grammar (species, <|on|> cluster-digits <|off|> ..., eos) with single- and multi-digit cluster
    # tokens, plus rows of arbitrary ids (bytes incl. non-ASCII continuation bytes, merged words, specials anywhere)
    rng = np.random.default_rng(0)
    ids_of = tok.convert_tokens_to_ids
    digit_tokens = sorted(i for t, i in bpe.items() if t.isdigit())
    prompt = ids_of(["<|startoftranscript|>", "<|en|>", "<|notimestamps|>"])
    rows = []
    for r in range(160):
        row = list(prompt) + [ids_of(SPECIES[rng.integers(len(SPECIES))])]
        for _ in range(rng.integers(0, 9)):
            on = int(rng.integers(0, 990))
            row += [ids_of("<|%d|>" % on)] + [int(digit_tokens[rng.integers(len(digit_tokens))]) for _ in range(rng.integers(1, 3))] + \
                   [ids_of("<|%d|>" % (on + int(rng.integers(1, 10))))]
        rows.append(row + [ids_of("<|endoftext|>")] * int(rng.integers(1, 3)))
    for r in range(80):
        rows.append([int(v) for v in rng.integers(0, len(tok), size=int(rng.integers(1, 40)))])
    cases = {"n_vocab": len(tok), "prompt": prompt, "eos_token_id": tok.eos_token_id,
             "rows": rows,
             "decoded": tok.batch_decode(rows, skip_special_tokens=False),
             "decoded_skip_special": tok.batch_decode(rows, skip_special_tokens=True)}
    with open(os.path.join(OUT, "decode_cases.json"), "w", encoding="utf-8") as f:
        json.dump(cases, f, ensure_ascii=False)
    print("wrote", OUT, "rows", len(rows), "vocab", len(tok))


if __name__ == "__main__":
    main()

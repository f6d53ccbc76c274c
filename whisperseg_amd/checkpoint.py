"""Checkpoint reader: an HF-style model directory -> tensors on demand.

Replaces what `WhisperForConditionalGeneration.from_pretrained(model_path)` reads at reference
model.py:633/636, i.e. whatever the reference's `save_pretrained` wrote (model.py:59-74).  With the
transformers default of 5 GB shards an fp32 whisperseg-large (6.2 GB) is saved as
`model-0000x-of-0000y.safetensors` + `model.safetensors.index.json`; older checkpoints use
`pytorch_model.bin` (optionally sharded with `pytorch_model.bin.index.json`).  All four layouts are read.

Tensors are fetched one at a time (`ckpt[name]`), so the host never holds a second full copy of the
model: safetensors shards are memory-mapped and sliced per tensor, `.bin` shards are loaded one shard
at a time (mmap when the file allows it) and dropped when the next shard is touched.
"""
import json
import os

import torch

SAFE_SINGLE, SAFE_INDEX = "model.safetensors", "model.safetensors.index.json"
BIN_SINGLE, BIN_INDEX = "pytorch_model.bin", "pytorch_model.bin.index.json"


def checkpoint_files(model_dir):
    """-> (kind, {tensor name: shard file}) with kind in {"safetensors", "bin"}; shard file None = single file,
    names unknown until opened.  Raises FileNotFoundError when the directory holds no HF weights."""
    for kind, single, index in (("safetensors", SAFE_SINGLE, SAFE_INDEX), ("bin", BIN_SINGLE, BIN_INDEX)):
        ipath = os.path.join(model_dir, index)
        if os.path.exists(ipath):
            with open(ipath) as f:
                weight_map = json.load(f)["weight_map"]
            missing = sorted({s for s in weight_map.values() if not os.path.exists(os.path.join(model_dir, s))})
            if missing:
                raise FileNotFoundError(f"{model_dir}: {index} names missing shard(s) {missing}")
            return kind, dict(weight_map)
        if os.path.exists(os.path.join(model_dir, single)):
            return kind, None
    raise FileNotFoundError(f"{model_dir} holds no HF weights ({SAFE_SINGLE}[.index.json] / {BIN_SINGLE}[.index.json]); "
                            "CTranslate2 model.bin is not supported")


def has_weights(model_dir):
    try:
        checkpoint_files(model_dir)
        return True
    except FileNotFoundError:
        return False


class LazyCheckpoint:
    """Mapping-like view of a (possibly sharded) checkpoint: `name in ckpt`, `ckpt.keys()`, `ckpt[name]` -> CPU tensor."""

    def __init__(self, model_dir):
        self.model_dir = model_dir
        self.kind, wm = checkpoint_files(model_dir)
        self._handles = {}        # safetensors: shard file -> open handle
        self._bin_cache = {}      # .bin shards: the two most recently used state dicts (LRU; prepare_weights interleaves
        self.non_mmap_loads = 0   # embedding / norm tensors with the layers, so one cached shard would thrash)
        if wm is None:
            single = SAFE_SINGLE if self.kind == "safetensors" else BIN_SINGLE
            wm = {k: single for k in self._shard_keys(single)}
        self.weight_map = wm

    def _safe(self, shard):
        h = self._handles.get(shard)
        if h is None:
            from safetensors import safe_open
            h = safe_open(os.path.join(self.model_dir, shard), framework="pt", device="cpu")
            self._handles[shard] = h
        return h

    def _bin(self, shard):
        if shard in self._bin_cache:
            self._bin_cache[shard] = self._bin_cache.pop(shard)      # most recently used last
            return self._bin_cache[shard]
        while len(self._bin_cache) >= 2:
            self._bin_cache.pop(next(iter(self._bin_cache)))
        path = os.path.join(self.model_dir, shard)
        try:
            sd = torch.load(path, map_location="cpu", weights_only=True, mmap=True)
        except (RuntimeError, ValueError):      # legacy (non-zipfile) serialisation cannot be mmapped: the whole shard is read
            self.non_mmap_loads += 1
            import warnings
            warnings.warn(f"{path}: legacy torch serialisation, loading the whole shard into host memory (no mmap)")
            sd = torch.load(path, map_location="cpu", weights_only=True)
        self._bin_cache[shard] = sd
        return sd

    def _shard_keys(self, shard):
        return list(self._safe(shard).keys()) if self.kind == "safetensors" else list(self._bin(shard).keys())

    def keys(self):
        return self.weight_map.keys()

    def __contains__(self, name):
        return name in self.weight_map

    def __len__(self):
        return len(self.weight_map)

    def shards(self):
        return sorted(set(self.weight_map.values()))

    def __getitem__(self, name):
        shard = self.weight_map[name]
        if self.kind == "safetensors":
            return self._safe(shard).get_tensor(name)
        return self._bin(shard)[name]

    def close(self):
        self._handles.clear()
        self._bin_cache = {}

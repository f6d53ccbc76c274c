"""Checkpoint reader (whisperseg_amd/checkpoint.py): every layout the reference's save_pretrained can produce
(reference model.py:59-74 -> HF save_pretrained: single / sharded safetensors, single / sharded .bin) yields the
same tensors, one at a time."""
import json
import os

import pytest
import torch
from safetensors.torch import load_file, save_file

from conftest import GOLDEN

TINY = os.path.join(GOLDEN, "tiny_model")


def tiny_sd():
    return load_file(os.path.join(TINY, "model.safetensors"))


def write_sharded(dst, sd, n_shards, kind):
    """Same files HF `save_pretrained(max_shard_size=...)` writes: shards in state-dict order + an index json."""
    names = list(sd)
    per = (len(names) + n_shards - 1) // n_shards
    weight_map = {}
    for i in range(n_shards):
        part = {k: sd[k].contiguous() for k in names[i * per:(i + 1) * per]}
        if kind == "safetensors":
            fname = f"model-{i + 1:05d}-of-{n_shards:05d}.safetensors"
            save_file(part, os.path.join(dst, fname))
        else:
            fname = f"pytorch_model-{i + 1:05d}-of-{n_shards:05d}.bin"
            torch.save(part, os.path.join(dst, fname))
        weight_map.update({k: fname for k in part})
    index = "model.safetensors.index.json" if kind == "safetensors" else "pytorch_model.bin.index.json"
    with open(os.path.join(dst, index), "w") as f:
        json.dump({"metadata": {"total_size": 0}, "weight_map": weight_map}, f)


@pytest.mark.parametrize("kind,n_shards", [("safetensors", 1), ("safetensors", 3), ("bin", 1), ("bin", 4)])
def test_every_layout_reads_the_same_tensors(tmp_path, kind, n_shards):
    from whisperseg_amd.checkpoint import LazyCheckpoint, has_weights
    sd = tiny_sd()
    if n_shards == 1:
        if kind == "safetensors":
            save_file(sd, str(tmp_path / "model.safetensors"))
        else:
            torch.save(sd, str(tmp_path / "pytorch_model.bin"))
    else:
        write_sharded(str(tmp_path), sd, n_shards, kind)
    assert has_weights(str(tmp_path))
    ck = LazyCheckpoint(str(tmp_path))
    assert ck.kind == kind and len(ck.shards()) == n_shards
    assert sorted(ck.keys()) == sorted(sd)
    for k in sd:      # bit-identical, whichever shard holds it
        assert torch.equal(ck[k], sd[k]), k
    # access pattern of prepare_weights: names of different shards interleaved
    assert torch.equal(ck["model.decoder.embed_tokens.weight"], sd["model.decoder.embed_tokens.weight"])
    assert torch.equal(ck["model.encoder.conv1.weight"], sd["model.encoder.conv1.weight"])
    ck.close()


def test_hf_save_pretrained_sharding_is_read(tmp_path):
    """The real writer: transformers' save_pretrained with a shard limit small enough to split the tiny model."""
    from transformers import WhisperConfig, WhisperForConditionalGeneration
    from whisperseg_amd.checkpoint import LazyCheckpoint
    with open(os.path.join(TINY, "config.json")) as f:
        cd = json.load(f)
    for k in ("total_spec_columns", "cluster_codebook", "default_segmentation_config", "model_type"):
        cd.pop(k)
    hf = WhisperForConditionalGeneration(WhisperConfig(**cd, suppress_tokens=None, begin_suppress_tokens=None))
    sd = {k: v.float() for k, v in tiny_sd().items()}
    sd["proj_out.weight"] = sd["model.decoder.embed_tokens.weight"]
    hf.load_state_dict(sd, strict=True)
    hf.save_pretrained(str(tmp_path), max_shard_size="1MB")
    assert os.path.exists(tmp_path / "model.safetensors.index.json")
    ck = LazyCheckpoint(str(tmp_path))
    assert len(ck.shards()) > 1
    for k, v in tiny_sd().items():
        assert torch.equal(ck[k].float(), v.float()), k


def test_missing_weights_and_missing_shard(tmp_path):
    from whisperseg_amd.checkpoint import LazyCheckpoint, has_weights
    assert not has_weights(str(tmp_path))
    with pytest.raises(FileNotFoundError):
        LazyCheckpoint(str(tmp_path))
    write_sharded(str(tmp_path), tiny_sd(), 2, "safetensors")
    os.remove(tmp_path / "model-00002-of-00002.safetensors")
    with pytest.raises(FileNotFoundError):
        LazyCheckpoint(str(tmp_path))


def test_prepare_weights_streams_from_a_lazy_checkpoint(tmp_path):
    """prepare_weights on a sharded LazyCheckpoint == prepare_weights on the in-memory state dict (CPU tensors here; the
    device path is the same code), and every tensor already has the model dtype."""
    from whisperseg_amd.checkpoint import LazyCheckpoint
    from whisperseg_amd.engine import geometry_from_config, prepare_weights
    with open(os.path.join(TINY, "config.json")) as f:
        cfg = json.load(f)
    geo = geometry_from_config(cfg)
    sd = tiny_sd()
    write_sharded(str(tmp_path), sd, 3, "safetensors")
    a = prepare_weights(LazyCheckpoint(str(tmp_path)), geo, torch.bfloat16, "cpu")
    b = prepare_weights(sd, geo, torch.bfloat16, "cpu")
    assert sorted(a) == sorted(b)
    for k in a:
        assert a[k].dtype == torch.bfloat16 and a[k].is_contiguous()
        assert torch.equal(a[k], b[k]), k

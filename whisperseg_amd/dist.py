"""Multi-GPU clip sharding: one process per GPU, torch.distributed (backend "nccl" == RCCL over xGMI).

The reference's only multi-GPU mechanism is a thread-per-device replica fan-out over contiguous chunks of
ceil(N / n_gpu) windows, re-joined in device order (reference model.py:169-189).  Windows never exchange
state, so the MI355X form is: every rank derives the same window table, decodes ITS contiguous shard
(same split rule), and the only collectives are
  * broadcast of the PCM of a recording from rank 0 (once per recording), and optionally of the weights,
  * all_gather of the generated token ids (int32 [per_rank, max_length] + lengths, <= 57 KB per rank),
after which rank order == window order, which is what parse_generation's index alignment needs
(reference model.py:221-222).  No all-reduce exists on this path.
"""
import os

import numpy as np
import torch
import torch.distributed as dist

from .windows import shard_bounds


def init_from_env(backend=None):
    """Initialise the default process group from torchrun's environment; returns (rank, world, local_rank)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    force = os.environ.get("WSEG_FORCE_DIST") == "1"      # exercise the RCCL collectives even at world size 1
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend=backend, rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def _single():
    """True when no collective is needed (WSEG_FORCE_DIST=1 keeps them on at world size 1, for validation)."""
    if not (dist.is_available() and dist.is_initialized()):
        return True
    return dist.get_world_size() == 1 and os.environ.get("WSEG_FORCE_DIST") != "1"


def my_shard(n_items, rank, world):
    """(lo, hi) of this rank under the reference's contiguous split; (n, n) when the rank gets nothing."""
    bounds = shard_bounds(n_items, world)
    return bounds[rank] if rank < len(bounds) else (n_items, n_items)


def broadcast_pcm(audio, device, src=0):
    """Rank `src` holds the recording (numpy float32); every rank returns it as a device tensor."""
    if _single():
        return torch.as_tensor(np.ascontiguousarray(audio, dtype=np.float32)).to(device)
    n = torch.zeros(1, dtype=torch.int64, device=device)
    if dist.get_rank() == src:
        n[0] = len(audio)
    dist.broadcast(n, src)
    if dist.get_rank() == src:
        buf = torch.as_tensor(np.ascontiguousarray(audio, dtype=np.float32)).to(device)
    else:
        buf = torch.empty(int(n.item()), dtype=torch.float32, device=device)
    if buf.numel():
        dist.broadcast(buf, src)
    return buf


def broadcast_weights(weights, src=0):
    """In-place broadcast of a prepared weight dict (same keys/shapes on every rank) from rank `src`.
    large bf16 = 3.08 GB: one pipelined ring broadcast is bounded by a single xGMI link (~153 GB/s)."""
    if not _single():
        for name in sorted(weights):
            dist.broadcast(weights[name], src)
    return weights


def gather_rows(tokens, lengths, n_total):
    """all_gather of per-rank results in rank (== window) order.

    tokens int32 [n_local, L], lengths int32 [n_local] for this rank's shard of n_total windows.
    Returns (tokens [n_total, L], lengths [n_total]) on every rank."""
    if _single():
        return tokens, lengths
    world = dist.get_world_size()
    per = int(np.ceil(n_total / world)) if n_total > 0 else 0
    L = tokens.shape[1]
    pad_t = torch.zeros((per, L), dtype=torch.int32, device=tokens.device)
    pad_l = torch.zeros((per,), dtype=torch.int32, device=tokens.device)
    pad_t[: tokens.shape[0]] = tokens
    pad_l[: lengths.shape[0]] = lengths
    all_t = torch.empty((world * per, L), dtype=torch.int32, device=tokens.device)
    all_l = torch.empty((world * per,), dtype=torch.int32, device=tokens.device)
    dist.all_gather_into_tensor(all_t, pad_t) if hasattr(dist, "all_gather_into_tensor") and tokens.is_cuda else \
        _all_gather_lists(all_t, pad_t, world)
    dist.all_gather_into_tensor(all_l, pad_l) if hasattr(dist, "all_gather_into_tensor") and tokens.is_cuda else \
        _all_gather_lists(all_l, pad_l, world)
    return all_t[:n_total], all_l[:n_total]


def _all_gather_lists(out, part, world):
    chunks = [torch.empty_like(part) for _ in range(world)]
    dist.all_gather(chunks, part)
    out.copy_(torch.cat(chunks, 0))


def segment_distributed(segmenter, audio, sr, decode_shard=None, **kwargs):
    """segment() of one recording with its windows sharded over the ranks of the default group.

    Every rank must call this; rank 0 supplies `audio` (other ranks may pass None).  Returns the
    prediction dict on every rank.  `decode_shard(sliced_shard, **gen) -> (tokens, lengths)` defaults to
    the segmenter's engine (tests inject a CPU stand-in to exercise the collectives under gloo)."""
    from . import postprocess
    from .audio_utils import get_n_fft_given_sr
    from .utils import RATIO_DECODING_TIME_STEP_TO_SPEC_TIME_STEP as RATIO
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    d = segmenter.default_segmentation_config
    min_frequency = kwargs.get("min_frequency", None)
    spec_time_step = kwargs.get("spec_time_step", None)
    if min_frequency is None:
        min_frequency = d.get("min_frequency", 0)
    if spec_time_step is None:
        spec_time_step = d.get("spec_time_step", 0.0025)
    num_trials = kwargs.get("num_trials", 1)
    device = segmenter.device_list[0]
    pcm = broadcast_pcm(audio, device)
    sliced = segmenter.sliced_features_from_device_pcm(pcm, sr, min_frequency, spec_time_step, num_trials, rank, world)
    n_total = sliced["n_total"]
    gen = dict(batch_size=kwargs.get("batch_size", 4), max_length=kwargs.get("max_length", 448),
               num_beams=kwargs.get("num_beams", 4), length_penalty=kwargs.get("length_penalty", 1.0))
    fn = decode_shard or segmenter.decode_shard_tokens
    tokens, lengths = fn(sliced["shard"], **gen)
    tokens, lengths = gather_rows(tokens, lengths, n_total)
    tokens, lengths = tokens.cpu().numpy(), lengths.cpu().numpy()
    texts = segmenter.tokens_to_texts(tokens, lengths)
    min_segment_length = kwargs.get("min_segment_length") or spec_time_step * RATIO
    eps = kwargs.get("eps") or spec_time_step * RATIO * 4
    frame = kwargs.get("time_per_frame_for_voting") or spec_time_step
    pred = segmenter.parse_generation(texts, sliced["table"], min_segment_length, pcm.numel() / sr, spec_time_step,
                                      num_trials, eps, frame, kwargs.get("consolidation_method", "clustering"))
    pred = postprocess.correct_fft_blur(pred, get_n_fft_given_sr(sr), sr)
    return postprocess.drop_consecutive_duplicates(pred)

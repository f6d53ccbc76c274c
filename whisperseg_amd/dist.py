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
            # as BYTES: the operand rows of the split-precision modes are int16 tensors, a dtype RCCL / NCCL has no name for
            # (ProcessGroupNCCL maps int8 / uint8 / int32 / int64 / half / float / double / bfloat16 only); a broadcast is a copy
            dist.broadcast(weights[name].view(torch.uint8), src)
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


def _decode_kwargs(kwargs):
    return dict(batch_size=kwargs.get("batch_size", 4), max_length=kwargs.get("max_length", 448),
                num_beams=kwargs.get("num_beams", 4), length_penalty=kwargs.get("length_penalty", 1.0),
                top_k=kwargs.get("top_k", 1), top_p=kwargs.get("top_p", 1.0))


def _resolve(segmenter, min_frequency, spec_time_step, min_segment_length, eps, frame):
    """segment()'s None-defaults (only None is a default: an explicit 0 stays 0)."""
    if hasattr(segmenter, "resolve_segmentation_params"):
        return segmenter.resolve_segmentation_params(min_frequency, spec_time_step, min_segment_length, eps, frame)
    from .utils import RATIO_DECODING_TIME_STEP_TO_SPEC_TIME_STEP as RATIO
    d = segmenter.default_segmentation_config
    min_frequency = d.get("min_frequency", 0) if min_frequency is None else min_frequency
    spec_time_step = d.get("spec_time_step", 0.0025) if spec_time_step is None else spec_time_step
    min_segment_length = spec_time_step * RATIO if min_segment_length is None else min_segment_length
    eps = spec_time_step * RATIO * 4 if eps is None else eps
    frame = spec_time_step if frame is None else frame
    return min_frequency, spec_time_step, min_segment_length, eps, frame


def segment_distributed(segmenter, audio, sr, decode_shard=None, **kwargs):
    """segment() of one recording with its windows sharded over the ranks of the default group.

    Every rank must call this; rank 0 supplies `audio` (other ranks may pass None).  Returns the
    prediction dict on every rank.  `decode_shard(sliced_shard, **gen) -> (tokens, lengths)` defaults to
    the segmenter's engine (tests inject a CPU stand-in to exercise the collectives under gloo).  Keyword arguments are
    segment()'s (model.py:397-470), with segment()'s None-defaults."""
    from . import postprocess
    from .audio_utils import get_n_fft_given_sr
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    min_frequency, spec_time_step, min_segment_length, eps, frame = _resolve(
        segmenter, kwargs.get("min_frequency"), kwargs.get("spec_time_step"), kwargs.get("min_segment_length"), kwargs.get("eps"),
        kwargs.get("time_per_frame_for_voting"))
    num_trials = kwargs.get("num_trials", 1)
    device = segmenter.device_list[0]
    pcm = broadcast_pcm(audio, device)
    sliced = segmenter.sliced_features_from_device_pcm(pcm, sr, min_frequency, spec_time_step, num_trials, rank, world)
    n_total = sliced["n_total"]
    fn = decode_shard or segmenter.decode_shard_tokens
    tokens, lengths = fn(sliced["shard"], **_decode_kwargs(kwargs))
    tokens, lengths = gather_rows(tokens, lengths, n_total)
    tokens, lengths = tokens.cpu().numpy(), lengths.cpu().numpy()
    texts = segmenter.tokens_to_texts(tokens, lengths)
    pred = segmenter.parse_generation(texts, sliced["table"], min_segment_length, pcm.numel() / sr, spec_time_step,
                                      num_trials, eps, frame, kwargs.get("consolidation_method", "clustering"))
    pred = postprocess.correct_fft_blur(pred, get_n_fft_given_sr(sr), sr)
    return postprocess.drop_consecutive_duplicates(pred)


def _broadcast_object(obj, src=0):
    if _single():
        return obj
    box = [obj]
    dist.broadcast_object_list(box, src)
    return box[0]


def _scatter_recordings(audios, meta, need, device, rank, world):
    """Rank 0 holds the recordings; every rank returns {recording index: device PCM} for the recordings in need[rank].
    One point-to-point message per destination rank (the recordings it needs, concatenated): a rank receives only the PCM
    its share of the pooled window list reads, not the whole batch."""
    mine = {}
    if rank == 0:
        for dst in range(1, world):
            if need[dst]:
                buf = np.concatenate([np.ascontiguousarray(audios[i], dtype=np.float32) for i in need[dst]])
                if buf.size:
                    dist.send(torch.from_numpy(buf).to(device), dst)
        for i in need[0]:
            mine[i] = torch.as_tensor(np.ascontiguousarray(audios[i], dtype=np.float32)).to(device)
    elif need[rank]:
        total = sum(meta[i]["n"] for i in need[rank])
        buf = torch.empty(total, dtype=torch.float32, device=device)
        if total:
            dist.recv(buf, 0)
        pos = 0
        for i in need[rank]:
            mine[i] = buf[pos:pos + meta[i]["n"]]
            pos += meta[i]["n"]
    return mine


def segment_batch_distributed(segmenter, audios, srs=None, decode_shard=None, **kwargs):
    """segment_batch() of a list of recordings — a folder, a multi-species batch (BASELINE configs[4]) — with the POOLED
    window list of all recordings partitioned over the ranks of the default group: the reference's fan-out shards whatever
    window list it is given into contiguous ceil(N / n_devices) chunks (model.py:169-189); here N is the window count of the
    whole batch and a chunk belongs to a process (one per GPU).

    Every rank must call this; rank 0 supplies `audios` / `srs` and the per-recording parameters (lists or scalars, exactly
    as SegmenterBase.segment_batch takes them); other ranks may pass None.  Exchange: (1) the recordings' metadata (lengths,
    rates, resolved parameters: a small pickled list, broadcast), (2) PCM point-to-point from rank 0 to the ranks whose
    windows read it, (3) all_gather of token ids + lengths.  Every rank then holds all tokens and parses every recording
    (host work, milliseconds); returns the list of prediction dicts on every rank, equal to per-file segment().
    `pool_windows` (default model.POOL_WINDOWS): recordings are processed in groups of at most ~pool_windows x world windows."""
    from . import postprocess
    from .audio_utils import get_n_fft_given_sr
    from .model import _per_item
    from .windows import window_table
    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    device = segmenter.device_list[0]
    meta = None
    if rank == 0:
        audios = list(audios)
        if srs is None:
            audios, sr_list = [a for a, _ in audios], [s for _, s in audios]
        elif isinstance(srs, (int, float)):
            sr_list = [srs] * len(audios)
        else:
            sr_list = list(srs)
        names = ("min_frequency", "spec_time_step", "min_segment_length", "eps", "time_per_frame_for_voting")
        per = [_per_item(kwargs.get(k)) for k in names] + [_per_item(kwargs.get("num_trials", 1)),
                                                            _per_item(kwargs.get("consolidation_method", "clustering"))]
        meta = []
        for audio, sr, mf, sts, msl, e, tpf, nt, method in zip(audios, sr_list, *per):
            mf, sts, msl, e, tpf = _resolve(segmenter, mf, sts, msl, e, tpf)
            meta.append(dict(n=int(len(audio)), sr=sr, min_frequency=mf, spec_time_step=sts, min_segment_length=msl, eps=e,
                             frame=tpf, num_trials=int(nt), method=method))
    # (the grouping cap travels with the metadata: non-zero ranks may pass None for every parameter, and ranks that derived different
    # groups would run different numbers of scatter / gather rounds and hang — ADVICE r04)
    meta, pool_windows = _broadcast_object((meta, kwargs.get("pool_windows") if rank == 0 else None))
    cols = segmenter.total_spec_columns
    counts = [len(window_table(m["n"], m["sr"], m["spec_time_step"], m["num_trials"], cols)) for m in meta]
    # Recordings are processed in GROUPS of at most ~pool_windows x world windows (every rank derives the same grouping from the
    # metadata): scatter, front-end, decode and gather run per group, so a rank never holds more than ~pool_windows windows of
    # log-mel features (320 KB each) however large the dataset is — evaluate() sends whole datasets through here.
    from .model import POOL_WINDOWS
    cap = max(1, int(pool_windows or POOL_WINDOWS)) * world
    groups, cur, cur_n = [], [], 0
    for i, c in enumerate(counts):
        if cur and cur_n + c > cap:
            groups.append(cur)
            cur, cur_n = [], 0
        cur.append(i)
        cur_n += c
    if cur:
        groups.append(cur)
    fn = decode_shard or segmenter.decode_shard_tokens
    out = []
    for members in groups:
        offs = np.concatenate([[0], np.cumsum([counts[i] for i in members])]).astype(np.int64)
        n_total = int(offs[-1])
        bounds = [my_shard(n_total, r, world) for r in range(world)]
        need = [[i for k, i in enumerate(members) if offs[k] < hi and offs[k + 1] > lo] for lo, hi in bounds]
        first = {i: int(offs[k]) for k, i in enumerate(members)}
        pcm = _scatter_recordings(audios, meta, need, device, rank, world) if not _single() else \
            {i: torch.as_tensor(np.ascontiguousarray(audios[i], dtype=np.float32)).to(device) for i in need[0]}
        lo, hi = bounds[rank]
        shard = []
        for i in need[rank]:
            m = meta[i]
            part = segmenter.sliced_features_from_device_pcm(pcm[i], m["sr"], m["min_frequency"], m["spec_time_step"], m["num_trials"],
                                                             window_range=(lo - first[i], hi - first[i]))
            shard += part["shard"]
        tokens, lengths = fn(shard, **_decode_kwargs(kwargs))
        del shard, pcm
        tokens, lengths = gather_rows(tokens, lengths, n_total)
        tokens, lengths = tokens.cpu().numpy(), lengths.cpu().numpy()
        texts = segmenter.tokens_to_texts(tokens, lengths)
        for k, i in enumerate(members):
            m = meta[i]
            rows = [(w.trial_id, w.offset_time, None, w.clip_seconds)
                    for w in window_table(m["n"], m["sr"], m["spec_time_step"], m["num_trials"], cols)]
            pred = segmenter.parse_generation(texts[int(offs[k]):int(offs[k + 1])], rows, m["min_segment_length"], m["n"] / m["sr"],
                                              m["spec_time_step"], m["num_trials"], m["eps"], m["frame"], m["method"])
            pred = postprocess.correct_fft_blur(pred, get_n_fft_given_sr(m["sr"]), m["sr"])
            out.append(postprocess.drop_consecutive_duplicates(pred))
    return out

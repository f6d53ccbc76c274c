"""N>1 path on CPU: world_size-2 gloo processes exercise whisperseg_amd.dist (PCM broadcast, contiguous
sharding, token all_gather, rank-order == window-order) with a deterministic stand-in for the GPU
decode.  Property: the sharded result equals the unsharded one, because shards are independent."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

L = 16


def fake_features(pcm, start, clip_len):
    """Stand-in 'features' of a window: the window's mean absolute amplitude (depends on content + position)."""
    lo, hi = max(start, 0), min(start + clip_len, pcm.numel())
    return float(pcm[lo:hi].abs().sum()) if hi > lo else 0.0


class FakeSegmenter:
    """Mirrors the hooks segment_distributed uses; decode is a pure function of the window content."""

    def __init__(self):
        from whisperseg_amd.model import SegmenterBase
        self.base = SegmenterBase()
        self.base.total_spec_columns = self.total_spec_columns = 1000
        self.base.cluster_codebook = {"a": 0, "b": 1, "c": 2}
        self.default_segmentation_config = self.base.default_segmentation_config = {"spec_time_step": 0.01, "min_frequency": 0}
        self.device_list = [torch.device("cpu")]
        self.parse_generation = self.base.parse_generation
        self.resolve_segmentation_params = self.base.resolve_segmentation_params
        self.pcm_samples_seen = 0

    def sliced_features_from_device_pcm(self, pcm, sr, min_frequency, spec_time_step, num_trials, rank=0, world=1,
                                        window_range=None):
        from whisperseg_amd.windows import shard_bounds, window_table
        table = window_table(int(pcm.numel()), sr, spec_time_step, num_trials, 1000)
        if window_range is not None:
            lo, hi = max(0, window_range[0]), min(len(table), window_range[1])
            hi = max(lo, hi)
        else:
            bounds = shard_bounds(len(table), world)
            lo, hi = bounds[rank] if rank < len(bounds) else (len(table), len(table))
        self.pcm_samples_seen += int(pcm.numel())
        clip_len = int(1000 * spec_time_step * sr)
        # the stand-in "features" also depend on the front-end parameters, as the real filterbank does
        shard = [(w.trial_id, w.offset_time, fake_features(pcm, w.start, clip_len) + 0.37 * (min_frequency > 0) + sr * 1e-6,
                  w.clip_seconds) for w in table[lo:hi]]
        rows = [(w.trial_id, w.offset_time, None, w.clip_seconds) for w in table]
        return {"table": rows, "shard": shard, "n_total": len(table), "lo": lo, "hi": hi}

    def decode_shard_tokens(self, shard, **gen):
        toks = torch.zeros((len(shard), L), dtype=torch.int32)
        lens = torch.zeros((len(shard),), dtype=torch.int32)
        for i, (_, _, f, _) in enumerate(shard):
            k = int(f * 7) % 5 + 1                       # number of segments, content dependent
            row = [9000]
            for j in range(k):
                on = (int(f * 13) + 40 * j) % 400
                row += [on, 100 + (int(f) + j) % 3, on + 20 + j]
            toks[i, : len(row)] = torch.tensor(row[:L], dtype=torch.int32)
            lens[i] = min(len(row), L)
        return toks, lens

    def tokens_to_texts(self, tokens, lengths):
        out = []
        for row, ln in zip(tokens, lengths):
            row = [int(t) for t in row[:ln]]
            txt = "<|unknown|>"
            for j in range(1, len(row) - 2, 3):
                txt += "<|%d|>%d<|%d|>" % (row[j], row[j + 1] - 100, row[j + 2])
            out.append(txt)
        return out


def make_audio():
    rng = np.random.default_rng(5)
    return (0.1 * rng.standard_normal(16000 * 47 + 321)).astype(np.float32)


def worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from whisperseg_amd import dist as wd
    wd.init_from_env(backend="gloo")
    seg = FakeSegmenter()
    audio = make_audio() if rank == 0 else None
    res = [wd.segment_distributed(seg, audio, 16000, spec_time_step=0.01, num_trials=nt, batch_size=2) for nt in (1, 3)]
    lo, hi = wd.my_shard(15, rank, world)
    q.put((rank, res, (lo, hi)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_sharded_equals_unsharded_gloo():
    from whisperseg_amd import dist as wd
    seg = FakeSegmenter()
    single = [wd.segment_distributed(seg, make_audio(), 16000, spec_time_step=0.01, num_trials=nt, batch_size=2) for nt in (1, 3)]
    assert len(single[0]["onset"]) > 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=150) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, res, _ in results:
        assert res == single, rank
    shards = sorted(r[2] for r in results)
    assert shards == [(0, 8), (8, 15)]


def test_gather_rows_single_process():
    from whisperseg_amd import dist as wd
    t = torch.arange(12, dtype=torch.int32).reshape(3, 4)
    ln = torch.tensor([4, 2, 3], dtype=torch.int32)
    a, b = wd.gather_rows(t, ln, 3)
    assert torch.equal(a, t) and torch.equal(b, ln)
    assert wd.my_shard(10, 3, 4) == (9, 10) and wd.my_shard(3, 5, 8) == (3, 3)


# ---- a clip BATCH partitioned over the ranks (BASELINE configs[4]: mixed 16 / 32 / 48 kHz, per-species parameters) ----------------
BATCH = [  # sr, seconds, spec_time_step, min_frequency, num_trials, min_segment_length, eps
    (16000, 23.4, 0.01, 0, 1, None, None),
    (32000, 7.9, 0.0025, 0, 3, 0.01, 0.02),
    (48000, 5.2, 0.0025, 1000, 2, 0.0, None),       # an explicit 0 must stay 0 (not "or default")
    (16000, 0.4, None, None, 1, None, None),        # checkpoint defaults
]


def make_batch():
    rng = np.random.default_rng(11)
    return [(0.1 * rng.standard_normal(int(sr * sec))).astype(np.float32) for sr, sec, *_ in BATCH]


def batch_kwargs():
    return dict(min_frequency=[b[3] for b in BATCH], spec_time_step=[b[2] for b in BATCH], num_trials=[b[4] for b in BATCH],
                min_segment_length=[b[5] for b in BATCH], eps=[b[6] for b in BATCH], batch_size=2, max_length=L)


def per_file_reference():
    """What per-file segment() gives: segment_distributed of each recording alone, single process."""
    from whisperseg_amd import dist as wd
    seg = FakeSegmenter()
    out = []
    for audio, (sr, _, sts, mf, nt, msl, eps) in zip(make_batch(), BATCH):
        out.append(wd.segment_distributed(seg, audio, sr, spec_time_step=sts, min_frequency=mf, num_trials=nt, min_segment_length=msl,
                                          eps=eps, batch_size=2, max_length=L))
    return out


def batch_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from whisperseg_amd import dist as wd
    wd.init_from_env(backend="gloo")
    seg = FakeSegmenter()
    audios = make_batch() if rank == 0 else None
    srs = [b[0] for b in BATCH] if rank == 0 else None
    res = wd.segment_batch_distributed(seg, audios, srs, **(batch_kwargs() if rank == 0 else dict(batch_size=2, max_length=L)))
    q.put((rank, res, seg.pcm_samples_seen))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("world", [2, 3])
def test_distributed_clip_batch_equals_per_file_segment_gloo(world):
    want = per_file_reference()
    assert sum(len(p["onset"]) for p in want) > 5 and len(want) == len(BATCH)
    from whisperseg_amd import dist as wd
    single = wd.segment_batch_distributed(FakeSegmenter(), make_batch(), [b[0] for b in BATCH], **batch_kwargs())
    assert single == want
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000 + world
    procs = [ctx.Process(target=batch_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=150) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    total = sum(len(a) for a in make_batch())
    for rank, res, seen in results:
        assert res == want, rank                       # every rank holds every recording's rows == per-file segment()
        assert seen < total                            # ... having received only the recordings its windows read


def test_per_recording_parameters_in_segment_batch_cpu():
    """SegmenterBase.segment_batch with per-recording lists == segment() per file (device stages stubbed)."""
    from whisperseg_amd.model import SegmenterBase
    from whisperseg_amd.windows import window_table

    class Stub(SegmenterBase):
        def __init__(self):
            super().__init__()
            self.total_spec_columns = 1000
            self.cluster_codebook = {"a": 0, "b": 1, "c": 2}
            self.default_segmentation_config = {"spec_time_step": 0.01, "min_frequency": 0}
            self.device_list = ["stub"]
            self.calls = 0

        def get_sliced_audios_features(self, audio, sr, min_frequency, spec_time_step, num_trials):
            clip_len = int(1000 * spec_time_step * sr)
            pcm = torch.from_numpy(np.asarray(audio))
            return [(w.trial_id, w.offset_time, fake_features(pcm, w.start, clip_len) + 0.37 * (min_frequency > 0) + sr * 1e-6, w.clip_seconds)
                    for w in window_table(len(audio), sr, spec_time_step, num_trials, 1000)]

        def generate_segment_text(self, sliced, *a, **k):
            self.calls += 1
            fs = FakeSegmenter()
            return fs.tokens_to_texts(*[t.numpy() for t in fs.decode_shard_tokens(sliced)])

    seg = Stub()
    kw = batch_kwargs()
    kw.pop("batch_size"); kw.pop("max_length")
    pooled = seg.segment_batch(make_batch(), [b[0] for b in BATCH], **kw)
    assert seg.calls == 1                                                # ONE pooled decode for the mixed-parameter batch
    for pred, audio, (sr, _, sts, mf, nt, msl, eps) in zip(pooled, make_batch(), BATCH):
        assert pred == seg.segment(audio, sr, min_frequency=mf, spec_time_step=sts, num_trials=nt, min_segment_length=msl, eps=eps)
    assert pooled == per_file_reference()
    with pytest.raises(ValueError):
        seg.segment_batch(make_batch(), [b[0] for b in BATCH], spec_time_step=[0.01, 0.01])      # list shorter than the batch


# ---- world 4: one rank receives ZERO windows, the batch holds an EMPTY recording, and grouping bounds the pooled list -------------
SMALL = [  # sr, seconds, spec_time_step, num_trials  -> 1 + 1 + 1 = 3 windows for 4 ranks
    (16000, 4.0, 0.01, 1),
    (16000, 0.0, 0.01, 1),          # an empty recording: ONE all-zero window ("this loop must be executed once even for zero
                                    # length audio", reference model.py:145-146)
    (16000, 7.5, 0.01, 1),
]


def make_small():
    rng = np.random.default_rng(23)
    return [(0.1 * rng.standard_normal(int(sr * sec))).astype(np.float32) for sr, sec, *_ in SMALL]


def small_kwargs(**extra):
    return dict(spec_time_step=[b[2] for b in SMALL], num_trials=[b[3] for b in SMALL], batch_size=2, max_length=L, **extra)


def small_worker(rank, world, port, q, pool_windows):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from whisperseg_amd import dist as wd
    wd.init_from_env(backend="gloo")
    seg = FakeSegmenter()
    audios = make_small() if rank == 0 else None
    srs = [b[0] for b in SMALL] if rank == 0 else None
    # only rank 0 names the grouping cap (the function's contract: other ranks may pass None for the parameters); the cap travels
    # with the broadcast metadata, else the ranks would form different groups and hang in the per-group collectives
    kw = small_kwargs(pool_windows=pool_windows) if rank == 0 else dict(batch_size=2, max_length=L)
    res = wd.segment_batch_distributed(seg, audios, srs, **kw)
    q.put((rank, res, seg.pcm_samples_seen))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(240)
@pytest.mark.parametrize("pool_windows", [None, 1])
def test_world4_idle_rank_and_empty_recording_gloo(pool_windows):
    """4 ranks, 3 windows, one of three recordings empty: rank 3 decodes nothing and still takes part in every collective;
    pool_windows=1 forces one GROUP per recording (the bounded-memory path evaluate() uses on whole datasets)."""
    from whisperseg_amd import dist as wd
    from whisperseg_amd.windows import window_table
    counts = [len(window_table(int(sr * sec), sr, sts, nt, 1000)) for sr, sec, sts, nt in SMALL]
    assert counts[1] == 1 and sum(counts) < 4
    seg = FakeSegmenter()
    want = [wd.segment_distributed(seg, a, sr, spec_time_step=sts, num_trials=nt, batch_size=2, max_length=L)
            for a, (sr, _, sts, nt) in zip(make_small(), SMALL)]
    assert len(want[1]["onset"]) == 0 and len(want[2]["onset"]) > 0
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000 + (7 if pool_windows else 0)
    procs = [ctx.Process(target=small_worker, args=(r, 4, port, q, pool_windows)) for r in range(4)]
    for p in procs:
        p.start()
    results = [q.get(timeout=200) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    seen = {}
    for rank, res, pcm_seen in results:
        assert res == want, rank
        seen[rank] = pcm_seen
    if pool_windows is None:
        assert seen[3] == 0                 # the rank without windows received no PCM at all


def eval_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from whisperseg_amd import dist as wd
    from whisperseg_amd.evaluate import evaluate
    from whisperseg_amd.model import SegmenterBase
    wd.init_from_env(backend="gloo")

    class Seg(SegmenterBase):            # our segment(), device stages stubbed; has decode_shard_tokens, i.e. COULD go distributed
        def __init__(self):
            super().__init__()
            self.fs = FakeSegmenter()
            self.total_spec_columns = 1000
            self.cluster_codebook = {"a": 0, "b": 1, "c": 2}
            self.default_segmentation_config = {"spec_time_step": 0.01, "min_frequency": 0}
            self.device_list = [torch.device("cpu")]
            self.sliced_features_from_device_pcm = self.fs.sliced_features_from_device_pcm
            self.decode_shard_tokens = self.fs.decode_shard_tokens
            self.tokens_to_texts = self.fs.tokens_to_texts

        def get_sliced_audios_features(self, audio, sr, min_frequency, spec_time_step, num_trials):
            pcm = torch.from_numpy(np.ascontiguousarray(audio, dtype=np.float32))
            return self.fs.sliced_features_from_device_pcm(pcm, sr, min_frequency, spec_time_step, num_trials)["shard"]

        def generate_segment_text(self, sliced, *a, **k):
            return self.fs.tokens_to_texts(*[t.numpy() for t in self.fs.decode_shard_tokens(sliced)])

    seg = Seg()
    audios = make_batch()[:2]
    labels = [dict(onset=[0.1], offset=[0.3], cluster=["a"], sr=BATCH[i][0], spec_time_step=BATCH[i][2]) for i in range(2)]
    out = None
    if rank == 0:                          # the training-loop idiom (reference train.py:250): ONLY rank 0 evaluates
        out = evaluate(audios, labels, seg, 2, L, 1)
    dist.barrier()
    both = evaluate(audios if rank == 0 else None, labels, seg, 2, L, 1, distributed=True)      # opt-in: every rank calls
    q.put((rank, out, both))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_rank0_only_evaluate_does_not_block_gloo():
    """an initialised process group must not turn evaluate() into a collective.  Rank 0 alone calls it (it would
    hang for ever in broadcast_object_list if it went distributed); with distributed=True both ranks call it and agree."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + os.getpid() % 2000
    procs = [ctx.Process(target=eval_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted((q.get(timeout=120) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert results[0][1] is not None and results[1][1] is None
    assert results[0][2] == results[1][2] == results[0][1]

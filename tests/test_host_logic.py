"""Product host logic (whisperseg_amd.windows / postprocess / model.SegmenterBase.segment) against golden
outputs of the reference's own segment() driven with the same generated texts.  No GPU: the device
stages are stubbed, everything else is the shipped code.  Comparison is EXACT float equality."""
import json
import os

import numpy as np
import pytest

import golden_inputs as GI
from whisperseg_amd import postprocess
from whisperseg_amd.model import SegmenterBase
from whisperseg_amd.windows import shard_bounds, window_table


def test_window_table_exact(golden_dir):
    with open(os.path.join(golden_dir, "windows.json")) as f:
        cases = json.load(f)
    for c in cases:
        rows = window_table(c["n"], c["sr"], c["sts"], c["trials"], 1000)
        assert [[w.trial_id, w.offset_time, w.clip_seconds] for w in rows] == c["table"], c


class StubSegmenter(SegmenterBase):
    """Device stages replaced: windows carry no features, texts come from the fixture."""

    def __init__(self, codebook, texts):
        super().__init__()
        self.total_spec_columns = 1000
        self.cluster_codebook = codebook
        self.texts = texts
        self.device_list = ["stub"]

    def get_sliced_audios_features(self, audio, sr, min_frequency, spec_time_step, num_trials):
        return [(w.trial_id, w.offset_time, None, w.clip_seconds)
                for w in window_table(len(audio), sr, spec_time_step, num_trials, self.total_spec_columns)]

    def generate_segment_text(self, sliced, *a, **k):
        assert len(sliced) == len(self.texts)
        return list(self.texts)


def load_cases(golden_dir):
    with open(os.path.join(golden_dir, "parse_cases.json")) as f:
        return json.load(f)


def test_segment_epilogue_exact(golden_dir):
    cases = load_cases(golden_dir)
    assert len(cases) >= 10
    for c in cases:
        audio = GI.signal("sine_noise", max(c["n"], 1), c["sr"], 21)[: c["n"]]
        seg = StubSegmenter(c["cluster_codebook"], c["texts"])
        got = seg.segment(audio, c["sr"], spec_time_step=c["sts"], **c["kwargs"])
        assert got == c["expected"], c["name"]


def test_dbscan_interval_equals_sklearn():
    """The banded DBSCAN used for large inputs yields sklearn's partition (same labels up to noise)."""
    from sklearn.cluster import DBSCAN
    rng = np.random.default_rng(0)
    for trial in range(20):
        n = int(rng.integers(1, 200))
        on = np.sort(rng.uniform(0, 20, n))
        pts = np.stack([on, on + rng.uniform(0.01, 0.5, n)], 1)
        pts = np.concatenate([pts, pts[: n // 2] + rng.normal(0, 0.01, (n // 2, 2))])
        eps, ms = float(rng.choice([0.01, 0.02, 0.08])), int(rng.integers(2, 4))
        dist = (np.abs(pts[:, None, 0] - pts[None, :, 0]) + np.abs(pts[:, None, 1] - pts[None, :, 1])) / 2
        want = DBSCAN(eps=eps, min_samples=ms, metric="precomputed").fit_predict(dist)
        got = postprocess.dbscan_interval(pts, eps, ms)
        assert np.array_equal(want, got)


def test_shard_bounds_is_reference_split():
    # ceil(N / n_dev) contiguous items per device, in order (reference model.py:172-175)
    assert shard_bounds(10, 4) == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert shard_bounds(8, 8) == [(i, i + 1) for i in range(8)]
    assert shard_bounds(3, 8) == [(0, 1), (1, 2), (2, 3)]
    assert shard_bounds(0, 4) == []
    assert shard_bounds(120, 8) == [(i * 15, i * 15 + 15) for i in range(8)]


def test_defaults_follow_reference():
    """segment() defaults (reference model.py:406-425): num_trials=1, num_beams=4, batch_size=4, max_length=448;
    min_segment_length = 2*sts, eps = 8*sts."""
    import inspect
    sig = inspect.signature(SegmenterBase.segment)
    d = {k: v.default for k, v in sig.parameters.items()}
    assert (d["num_trials"], d["num_beams"], d["batch_size"], d["max_length"], d["top_k"], d["top_p"], d["length_penalty"]) \
        == (1, 4, 4, 448, 1, 1.0, 1.0)
    assert d["consolidation_method"] == "clustering" and d["status_monitor"] is None


def test_scoring_helpers():
    seg = SegmenterBase()
    pred = {"onset": [0.1, 1.0, 2.0], "offset": [0.5, 1.5, 2.5], "cluster": ["a", "b", "a"]}
    lab = {"onset": [0.105, 1.2, 2.0], "offset": [0.5, 1.7, 2.5], "cluster": ["a", "b", "b"]}
    # expected values recorded from the reference's SegmenterBase.segment_score / frame_score on these inputs
    import copy
    got = [float(v) for v in seg.segment_score(copy.deepcopy(pred), copy.deepcopy(lab), tolerance=0.01)]
    assert got == [1.0, 3.0, 3.0, 0.3333333333333333, 0.3333333333333333, 0.3333333333333333]
    got = [float(v) for v in seg.frame_score(copy.deepcopy(pred), copy.deepcopy(lab), time_per_frame_for_scoring=0.001)]
    assert got == [695.0, 1400.0, 1395.0, 0.49642857142857144, 0.4982078853046595, 0.4973166368515206]
    got = [float(v) for v in seg.frame_score(copy.deepcopy(pred), copy.deepcopy(lab), target_cluster="b", time_per_frame_for_scoring=0.001)]
    assert got == [300.0, 500.0, 1000.0, 0.6, 0.3, 0.4]


def test_evaluate_accumulates_like_reference(tmp_path):
    """whisperseg_amd.evaluate.evaluate with a stub segmenter: pooled counts and F1 (reference evaluate.py:9-51)."""
    import json as _json
    from whisperseg_amd.evaluate import evaluate, get_audio_and_label_paths, read_label

    class Stub(SegmenterBase):
        def __init__(self, preds):
            super().__init__()
            self.preds = iter(preds)

        def segment(self, audio, sr, **kw):
            return next(self.preds)

    labels = [{"onset": [0.1, 1.0], "offset": [0.5, 1.5], "cluster": ["a", "b"], "sr": 16000},
              {"onset": [0.2], "offset": [0.9], "cluster": ["a"], "sr": 16000, "spec_time_step": 0.01}]
    preds = [{"onset": [0.1, 1.2], "offset": [0.5, 1.5], "cluster": ["a", "b"]}, {"onset": [0.2], "offset": [0.9], "cluster": ["a"]}]
    res = evaluate([np.zeros(10), np.zeros(10)], labels, Stub(preds), 8, 448, 1)
    assert res["segment_wise"][:3] == [2, 3, 3] and abs(res["segment_wise"][5] - 2 / 3) < 1e-12
    tp, n_pred, n_lab = res["frame_wise"][:3]
    assert (tp, n_pred, n_lab) == (400 + 300 + 700, 400 + 300 + 700, 400 + 500 + 700)
    (tmp_path / "x.wav").write_bytes(b"")
    (tmp_path / "x.json").write_text(_json.dumps({"onset": [0], "offset": [1]}))
    (tmp_path / "y.wav").write_bytes(b"")
    (tmp_path / "y.csv").write_text("onset,offset,cluster\n0.5,0.75,3\n")
    a, l = get_audio_and_label_paths(str(tmp_path))
    assert sorted(os.path.basename(p) for p in l) == ["x.json", "y.csv"]
    lab = read_label(str(tmp_path / "y.csv"))
    assert lab["onset"] == [0.5] and lab["cluster"] == ["3"] and lab["species"] == "unknown"
    assert read_label(str(tmp_path / "x.json"))["cluster"] == ["Vocal"]


def test_in_process_device_fan_out_two_stub_devices():
    """a-6 (reference model.py:169-189): the thread-per-device fan-out with TWO devices in one process — contiguous
    ceil(N / n_devices) shards, one thread per device with its own replica index, results re-joined in device order, a worker's
    exception re-raised (the reference loses it).  Device stages are stubs: no GPU has ever run this with two devices."""
    import threading
    import time
    import pytest
    from whisperseg_amd.model import SegmenterBase

    class TwoDevices(SegmenterBase):
        def __init__(self, fail_on=None):
            super().__init__()
            self.device_list = ["stub:0", "stub:1"]
            self.seen = {}
            self.fail_on = fail_on
            self.barrier = threading.Barrier(2, timeout=20)

        def generate_segment_text_core(self, sliced, batch_size, max_length, num_beams, top_k, top_p, length_penalty,
                                       generated_texts_dict, thread_id, status_monitor=None):
            self.barrier.wait()                                  # both device threads are alive at the same time
            self.seen[thread_id] = (threading.get_ident(), [s[0] for s in sliced], status_monitor is not None)
            if thread_id == 0:
                time.sleep(0.05)                                 # device 0 finishes LAST: order must come from the device index
            if self.fail_on == thread_id:
                raise RuntimeError("device %d failed" % thread_id)
            generated_texts_dict[thread_id] = ["w%d@%d" % (s[0], thread_id) for s in sliced]

    seg = TwoDevices()
    windows = [(i, 0.0, None, 1.0) for i in range(7)]
    texts = seg.generate_segment_text(windows, 4, 448, 4, status_monitor={})
    assert texts == ["w0@0", "w1@0", "w2@0", "w3@0", "w4@1", "w5@1", "w6@1"]          # ceil(7 / 2) = 4 | 3, device order
    assert seg.seen[0][0] != seg.seen[1][0]                                            # two threads
    assert seg.seen[0][2] and not seg.seen[1][2]                                       # the monitor goes to device 0 only
    for bad in (0, 1):
        with pytest.raises(RuntimeError, match="device %d failed" % bad):
            TwoDevices(fail_on=bad).generate_segment_text(windows, 4, 448, 4)
    # fewer windows than devices: the second device gets an empty shard or no thread at all, never an index error
    one = TwoDevices()
    one.barrier = threading.Barrier(1)
    assert one.generate_segment_text(windows[:1], 4, 448, 4) == ["w0@0"]

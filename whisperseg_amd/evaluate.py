"""Dataset-level evaluation — counterpart of reference evaluate.py:9-84 (segment every labelled recording, pool
segment-wise and frame-wise TP / P / R / F1).  Label files follow reference datautils.py:19-44 (json or csv with
onset / offset [/ cluster], optional sr / min_frequency / spec_time_step)."""
import csv
import json
import os

from .wavio import load_wav


def read_label(label_path, default_config={}, ignore_cluster=False):
    """reference datautils.py:19-44"""
    if label_path.endswith(".json"):
        with open(label_path) as f:
            label = json.load(f)
    elif label_path.endswith(".csv"):
        with open(label_path, newline="") as f:
            rows = list(csv.DictReader(f))
        label = {k: [r[k] for r in rows] for k in (rows[0].keys() if rows else [])}
        for k in ("onset", "offset"):
            if k in label:
                label[k] = [float(v) for v in label[k]]
    else:
        raise AssertionError("Unsupported file format!")
    assert "onset" in label and "offset" in label
    if "cluster" not in label:
        label["cluster"] = ["Vocal"] * len(label["onset"])
    label["cluster"] = list(map(str, label["cluster"]))
    for k in default_config:
        label.setdefault(k, default_config[k])
    label["species"] = "unknown"
    if ignore_cluster:
        label["cluster"] = ["Vocal"] * len(label["cluster"])
    return label


def get_audio_and_label_paths(folder):
    """reference datautils.py:46-58: every .wav with a sibling .json (preferred) or .csv."""
    audio_paths, label_paths = [], []
    for fname in os.listdir(folder):
        if not fname.endswith(".wav"):
            continue
        stem = os.path.join(folder, fname[:-4])
        for ext in (".json", ".csv"):
            if os.path.exists(stem + ext):
                audio_paths.append(stem + ".wav")
                label_paths.append(stem + ext)
                break
    return audio_paths, label_paths


def _prf(tp, n_pred, n_label):
    precision = tp / max(n_pred, 1e-12)
    recall = tp / max(n_label, 1e-12)
    return [tp, n_pred, n_label, precision, recall, 2 / (1 / max(precision, 1e-12) + 1 / max(recall, 1e-12))]


def evaluate(audio_list, label_list, segmenter, batch_size, max_length, num_trials, num_beams=4, target_cluster=None,
             distributed=None):
    """reference evaluate.py:9-51 -> {"segment_wise": [TP, P_pred, P_label, precision, recall, f1], "frame_wise": [...]}

    The reference segments file by file with each label's own sr / min_frequency / spec_time_step (evaluate.py:15-24).  Here
    all recordings go through ONE pooled decode with those per-recording parameters (SegmenterBase.segment_batch), which yields
    the per-file predictions of segment() (pooling only changes the batch a window is decoded in).  A segmenter without
    segment_batch (any object with the reference's interface) is driven file by file as upstream.

    `distributed` is OPT-IN (default: $WSEG_EVAL_DISTRIBUTED == "1", else False): the reference calls evaluate() from inside
    its training loop (train.py:250), where the usual `if rank == 0: evaluate(...)` under DDP / torchrun must stay a purely
    local call — an initialised process group alone never makes this function a collective.  With distributed=True EVERY rank
    of the default group must call evaluate(); the clip batch is then sharded over the ranks
    (dist.segment_batch_distributed; rank 0's audio_list is used)."""
    from . import dist as wdist
    kw = dict(min_frequency=[label.get("min_frequency", None) for label in label_list],
              spec_time_step=[label.get("spec_time_step", None) for label in label_list],
              max_length=max_length, batch_size=batch_size, num_trials=num_trials, num_beams=num_beams)
    srs = [label["sr"] for label in label_list]
    from .model import SegmenterBase
    ours = getattr(type(segmenter), "segment", None) is SegmenterBase.segment      # not a subclass with its own segment()
    if distributed is None:
        distributed = os.environ.get("WSEG_EVAL_DISTRIBUTED") == "1"
    if distributed and ours and not wdist._single() and hasattr(segmenter, "decode_shard_tokens"):
        predictions = wdist.segment_batch_distributed(segmenter, audio_list, srs, **kw)
    elif ours and hasattr(segmenter, "segment_batch"):
        predictions = segmenter.segment_batch(audio_list, srs, **kw)
    else:
        predictions = [segmenter.segment(audio, sr=sr, min_frequency=mf, spec_time_step=sts, max_length=max_length,
                                         batch_size=batch_size, num_trials=num_trials, num_beams=num_beams)
                       for audio, sr, mf, sts in zip(audio_list, srs, kw["min_frequency"], kw["spec_time_step"])]
    seg_tot, frame_tot = [0, 0, 0], [0, 0, 0]
    for prediction, label in zip(predictions, label_list):
        for tot, scores in ((seg_tot, segmenter.segment_score(prediction, label, target_cluster=target_cluster)[:3]),
                            (frame_tot, segmenter.frame_score(prediction, label, target_cluster=target_cluster)[:3])):
            for i in range(3):
                tot[i] += scores[i]
    return {"segment_wise": _prf(*seg_tot), "frame_wise": _prf(*frame_tot)}


def evaluate_dataset(dataset_folder, model_path, num_trials, max_length=448, num_beams=4, batch_size=8, **kwargs):
    """reference evaluate.py:53-84.  Recordings whose native rate differs from the label's `sr` are resampled (the
    reference does it inside librosa.load) with whisperseg_amd.resample."""
    from .model import WhisperSegmenter, WhisperSegmenterFast
    audio_list, label_list = [], []
    audio_paths, label_paths = get_audio_and_label_paths(dataset_folder)
    for audio_path, label_path in zip(audio_paths, label_paths):
        label = read_label(label_path)
        audio, sr = load_wav(audio_path)
        want = label.get("sr", None)
        if want is not None and int(want) != sr:      # librosa.load(path, sr=label sr) upstream: resample on the GPU
            from .resample import resample
            audio, sr = resample(audio, sr, int(want)), int(want)
        label["sr"] = sr
        audio_list.append(audio)
        label_list.append(label)
    try:
        segmenter = WhisperSegmenterFast(model_path=model_path, device="cuda")
    except Exception:
        segmenter = WhisperSegmenter(model_path=model_path, device="cuda")
    res = evaluate(audio_list, label_list, segmenter, batch_size, max_length, num_trials, num_beams, target_cluster=None)
    names = ["N-true-positive", "N-positive-in-prediction", "N-positive-in-ground-truth", "precision", "recall", "F1"]
    return {"segment_wise_scores": dict(zip(names, res["segment_wise"])),
            "frame_wise_scores": dict(zip(names, res["frame_wise"]))}

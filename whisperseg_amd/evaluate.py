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


def evaluate(audio_list, label_list, segmenter, batch_size, max_length, num_trials, num_beams=4, target_cluster=None):
    """reference evaluate.py:9-51 -> {"segment_wise": [TP, P_pred, P_label, precision, recall, f1], "frame_wise": [...]}"""
    seg_tot, frame_tot = [0, 0, 0], [0, 0, 0]
    for audio, label in zip(audio_list, label_list):
        prediction = segmenter.segment(audio, sr=label["sr"], min_frequency=label.get("min_frequency", None),
                                       spec_time_step=label.get("spec_time_step", None), max_length=max_length,
                                       batch_size=batch_size, num_trials=num_trials, num_beams=num_beams)
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

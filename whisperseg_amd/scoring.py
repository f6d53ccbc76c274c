"""Segment-wise and frame-wise precision / recall / F1 (reference model.py:474-569), used by evaluate-style
callers.  Pure host arithmetic."""
import numpy as np


def _count_matches(pred_rows, label_rows, tolerance):
    """Greedy one-to-one matching in prediction order (model.py:474-491)."""
    label_rows = list(label_rows)
    hits = 0
    for p_on, p_off, p_c in pred_rows:
        for i, (l_on, l_off, l_c) in enumerate(label_rows):
            if np.abs(p_on - l_on) <= tolerance and np.abs(p_off - l_off) <= tolerance and p_c == l_c:
                hits += 1
                label_rows.pop(i)
                break
    return hits


def _prf(tp, n_pred, n_label):
    precision = tp / max(n_pred, 1e-12)
    recall = tp / max(n_label, 1e-12)
    f1 = 2 / (1 / max(precision, 1e-12) + 1 / max(recall, 1e-12))
    return precision, recall, f1


def _rows(seg, target_cluster):
    return [[seg["onset"][i], seg["offset"][i], str(seg["cluster"][i])] for i in range(len(seg["onset"]))
            if target_cluster is None or str(target_cluster) == str(seg["cluster"][i])]


def segment_score(prediction, label, target_cluster=None, tolerance=0.01):
    """model.py:493-516 -> (TP, P_pred, P_label, precision, recall, f1)."""
    pred, lab = _rows(prediction, target_cluster), _rows(label, target_cluster)
    if target_cluster is not None and len(lab) == 0:
        print("Warning: the specified target cluster '%s' does not exist in the ground-truth labels." % str(target_cluster))
    n_pred, n_lab = len(pred), len(lab)
    tp = _count_matches(pred, lab, tolerance)
    return (tp, n_pred, n_lab) + _prf(tp, n_pred, n_lab)


def frame_score(prediction, label, target_cluster=None, time_per_frame_for_scoring=0.001):
    """model.py:518-569 -> (TP, P_pred, P_label, precision, recall, f1) on a frame raster."""
    prediction["cluster"] = list(map(str, prediction["cluster"]))
    label["cluster"] = list(map(str, label["cluster"]))
    ids = {}
    for c in list(prediction["cluster"]) + list(label["cluster"]):
        ids.setdefault(c, len(ids))
    stamps = list(prediction["onset"]) + list(prediction["offset"]) + list(label["onset"]) + list(label["offset"])
    t_max = np.max(stamps) if stamps else 1.0
    n = int(np.round(t_max / time_per_frame_for_scoring)) + 1

    def raster(seg):
        g = np.ones(n) * -1
        for on, off, c in zip(seg["onset"], seg["offset"], seg["cluster"]):
            g[int(np.round(on / time_per_frame_for_scoring)): int(np.round(off / time_per_frame_for_scoring))] = ids[c]
        return g

    gp, gl = raster(prediction), raster(label)
    if target_cluster is None:
        tp = np.logical_and(gl != -1, gp == gl).sum()
        n_pred, n_lab = (gp != -1).sum(), (gl != -1).sum()
    else:
        cid = ids[target_cluster]
        tp = np.logical_and(gl == cid, gp == gl).sum()
        n_pred, n_lab = (gp == cid).sum(), (gl == cid).sum()
    return (tp, n_pred, n_lab) + _prf(tp, n_pred, n_lab)

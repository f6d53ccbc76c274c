"""CPU epilogue of segmentation: generated text -> segments -> stitched / filtered / consolidated rows.

Restates reference model.py:191-394 and :439-468 with identical float arithmetic (exact `==` stitching,
rounding to 3 decimals BEFORE the FFT-blur correction, first-max majority vote, ...).  Costs
microseconds per window; it stays on the host.
"""
import re

import numpy as np

from .utils import RATIO_DECODING_TIME_STEP_TO_SPEC_TIME_STEP

# onset token, cluster digits, offset token, adjacent with nothing between (reference model.py:120)
SEGMENT_PATTERN = re.compile(r"<\|([0-9]+)\|>(\d+?)<\|([0-9]+)\|>")


def _empty():
    return {"onset": [], "offset": [], "cluster": []}


def extract_segments(text, spec_time_step, cluster_codebook, matcher=SEGMENT_PATTERN):
    """model.py:191-207: [[onset_s, offset_s, cluster_name]] relative to the window start."""
    names = {v: k for k, v in cluster_codebook.items()}
    rows = []
    for on_txt, cid_txt, off_txt in matcher.findall(text):
        # NB: the reference multiplies int * spec_time_step first, then * RATIO (left to right)
        onset = int(on_txt) * spec_time_step * RATIO_DECODING_TIME_STEP_TO_SPEC_TIME_STEP
        offset = int(off_txt) * spec_time_step * RATIO_DECODING_TIME_STEP_TO_SPEC_TIME_STEP
        cid = int(cid_txt)
        if cid not in names or offset - onset <= 0:
            continue
        rows.append([onset, offset, names[cid]])
    return rows


def stitch_trial(clips):
    """model.py:238-248: concatenate the per-window lists of one trial, merging a segment that ends
    exactly where the next window's first segment starts (same cluster, exact float equality)."""
    merged = []
    for clip in clips:
        if merged and clip and merged[-1][1] == clip[0][0] and merged[-1][2] == clip[0][2]:
            merged[-1][1] = clip[0][1]
            clip = clip[1:]
        merged += clip
    return merged


def custom_distance(a, b):
    """model.py:285-288"""
    return (abs(a[0] - b[0]) + abs(a[1] - b[1])) / 2


def _dbscan_labels(points, eps, min_samples, dense_limit=4000):
    """DBSCAN over the precomputed interval distance.  Small inputs go through scikit-learn exactly as
    the reference does (model.py:305-309); large ones use a banded neighbour search (only intervals with
    |d onset| <= 2 eps can be within eps) with scikit-learn's expansion order, which yields the same
    partition without the O(n^2) matrix."""
    n = len(points)
    pts = np.asarray(points, dtype=np.float64).reshape(n, 2)
    if n <= dense_limit:
        try:
            from sklearn.cluster import DBSCAN
            dist = (np.abs(pts[:, None, 0] - pts[None, :, 0]) + np.abs(pts[:, None, 1] - pts[None, :, 1])) / 2
            return DBSCAN(eps=eps, min_samples=min_samples, metric="precomputed").fit_predict(dist)
        except ImportError:
            pass
    return dbscan_interval(pts, eps, min_samples)


def dbscan_interval(pts, eps, min_samples):
    """Own DBSCAN (same visiting order as sklearn's dbscan_inner: seeds in index order, LIFO expansion)."""
    n = len(pts)
    order = np.argsort(pts[:, 0], kind="stable")
    onsets_sorted = pts[order, 0]
    neigh = [None] * n
    for i in range(n):
        lo = np.searchsorted(onsets_sorted, pts[i, 0] - 2 * eps, side="left")
        hi = np.searchsorted(onsets_sorted, pts[i, 0] + 2 * eps, side="right")
        cand = np.sort(order[lo:hi])
        d = (np.abs(pts[cand, 0] - pts[i, 0]) + np.abs(pts[cand, 1] - pts[i, 1])) / 2
        neigh[i] = cand[d <= eps]
    core = np.array([len(nb) >= min_samples for nb in neigh], dtype=bool)
    labels = np.full(n, -1, dtype=np.intp)
    label = 0
    for seed in range(n):
        if labels[seed] != -1 or not core[seed]:
            continue
        stack, i = [], seed
        while True:
            if labels[i] == -1:
                labels[i] = label
                if core[i]:
                    for v in neigh[i]:
                        if labels[v] == -1:
                            stack.append(v)
            if not stack:
                break
            i = stack.pop()
        label += 1
    return labels


def consolidate_by_clustering(trials, eps, min_samples):
    """model.py:291-337: pool every trial's segments, DBSCAN them, average each cluster's boundaries and
    take the majority name (first maximum in insertion order)."""
    pooled = [(on, off, name) for t in trials for on, off, name in zip(t["onset"], t["offset"], t["cluster"])]
    if not pooled:
        return _empty()
    labels = _dbscan_labels([[on, off] for on, off, _ in pooled], eps, min_samples)
    merged = []
    for lab in set(labels.tolist()):
        if lab == -1:
            continue
        group = [seg for seg, l in zip(pooled, labels) if l == lab]
        if not group:
            continue
        votes = {}
        for _, _, name in group:
            votes[name] = votes.get(name, 0) + 1
        best = sorted(votes.items(), key=lambda kv: -kv[1])[0][0]
        merged.append((np.mean([g[0] for g in group]), np.mean([g[1] for g in group]), best))
    merged.sort(key=lambda r: r[0])
    return {"onset": [r[0] for r in merged], "offset": [r[1] for r in merged], "cluster": [r[2] for r in merged]}


def consolidate_by_voting(trials, frame, cluster_codebook):
    """model.py:339-394: rasterise each trial on a frame grid, per-frame mode across trials, run-length
    decode back to segments."""
    from scipy.stats import mode
    stamps = []
    for t in trials:
        stamps += list(t["onset"])
        stamps += list(t["offset"])
    if len(stamps) == 0 or len(stamps) % 2 != 0:
        return _empty()
    t_min, t_max = np.min(stamps), np.max(stamps)
    n_frames = int(np.round((t_max - t_min) / frame))
    grids = []
    for t in trials:
        grid = np.ones(n_frames) * -1
        for on, off, name in zip(t["onset"], t["offset"], t["cluster"]):
            a = int(np.round((on - t_min) / frame))
            b = int(np.round((off - t_min) / frame))
            grid[a:b] = cluster_codebook[name]
        grids.append(grid)
    voted, _ = mode(np.asarray(grids), axis=0)
    voted = np.asarray(voted).reshape(-1)
    right = np.array(voted.tolist() + [-1])
    left = np.array([-1] + voted.tolist())
    edges = np.argwhere(right - left != 0)[:, 0]
    names = {v: k for k, v in cluster_codebook.items()}
    out = _empty()
    for a, b in zip(edges[:-1], edges[1:]):
        cid = int(np.round(np.mean(voted[a:b])))
        if cid == -1:
            continue
        out["onset"].append(a * frame + t_min)
        out["offset"].append(b * frame + t_min)
        out["cluster"].append(names[cid])
    return out


def parse_generation(texts, windows, cluster_codebook, min_segment_length, audio_duration, spec_time_step,
                     num_trials, eps, time_per_frame_for_voting, consolidation_method, precision_bits=3,
                     matcher=SEGMENT_PATTERN):
    """model.py:210-281.  `windows[i]` = (trial_id, offset_time, <features>, clip_seconds) of texts[i]."""
    per_trial = {}
    for text, win in zip(texts, windows):
        trial_id, offset_time = win[0], win[1]
        rows = extract_segments(text, spec_time_step, cluster_codebook, matcher)
        for row in rows:
            row[0] += offset_time
            row[1] += offset_time
        per_trial.setdefault(trial_id, []).append(rows)
    results = []
    for trial_id in per_trial:
        rows = stitch_trial(per_trial[trial_id])
        for row in rows:
            row[0] = max(0, row[0])
            row[1] = min(row[1], audio_duration)
        rows = sorted(rows, key=lambda r: r[0])
        rows = [r for r in rows if r[1] - r[0] >= min_segment_length]
        results.append({"onset": [r[0] for r in rows], "offset": [r[1] for r in rows], "cluster": [r[2] for r in rows]})
    if num_trials == 1:
        final = results[0]
    elif consolidation_method == "clustering":
        final = consolidate_by_clustering(results, eps, max(2, int(np.ceil(num_trials * 0.5))))
    else:
        final = consolidate_by_voting(results, time_per_frame_for_voting, cluster_codebook)
    final["onset"] = [float(np.round(t, precision_bits)) for t in final["onset"]]
    final["offset"] = [float(np.round(t, precision_bits)) for t in final["offset"]]
    return final


def correct_fft_blur(prediction, n_fft, sr):
    """model.py:439-455: shrink every segment by half an FFT window on both sides; a segment that would
    invert collapses to its midpoint."""
    delta = n_fft / 2 / sr
    ons, offs = [], []
    for on, off in zip(prediction["onset"], prediction["offset"]):
        a, b = on + delta, off - delta
        if a > b:
            a = b = (on + off) / 2
        ons.append(a)
        offs.append(b)
    prediction["onset"], prediction["offset"] = ons, offs
    return prediction


def drop_consecutive_duplicates(prediction):
    """model.py:457-468: after sorting by onset, drop rows identical to the previous kept row."""
    if len(prediction["onset"]) == 0:
        return prediction
    kept = []
    for row in sorted(zip(prediction["onset"], prediction["offset"], prediction["cluster"]), key=lambda r: r[0]):
        if not kept or row[0] != kept[-1][0] or row[1] != kept[-1][1] or row[2] != kept[-1][2]:
            kept.append(row)
    prediction["onset"] = [r[0] for r in kept]
    prediction["offset"] = [r[1] for r in kept]
    prediction["cluster"] = [r[2] for r in kept]
    return prediction

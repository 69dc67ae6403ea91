"""CPU restatement of the epoch-tail metrics (test infrastructure).

Follows reference `src/eoe/training/ad_trainer.py:452-455,516-522`: `sklearn.metrics.roc_curve` + `auc`
(trapezoid) and `average_precision_score`.  Third-party arithmetic: scikit-learn >= 1.0.2
(`src/requirements.txt`).  Restated as the tie-aware rank statistic (Mann-Whitney U), which equals the
trapezoidal area under the ROC curve, and as the step-wise precision-recall sum.
"""
import numpy as np


def roc_auc(labels: np.ndarray, scores: np.ndarray) -> float:
    labels = np.asarray(labels).astype(np.int64).ravel()
    scores = np.asarray(scores, dtype=np.float64).ravel()
    pos = labels == 1
    n_pos = int(pos.sum())
    n_neg = labels.size - n_pos
    if n_pos == 0 or n_neg == 0:
        return float("nan")
    order = np.argsort(scores, kind="mergesort")
    s = scores[order]
    # average ranks over ties
    ranks = np.empty(s.size, dtype=np.float64)
    i = 0
    n = s.size
    boundaries = np.flatnonzero(np.diff(s) != 0) + 1
    starts = np.concatenate([[0], boundaries])
    ends = np.concatenate([boundaries, [n]])
    for a, b in zip(starts, ends):
        ranks[a:b] = 0.5 * (a + b - 1) + 1.0
    r = np.empty(n, dtype=np.float64)
    r[order] = ranks
    u = r[pos].sum() - n_pos * (n_pos + 1) / 2.0
    return float(u / (n_pos * n_neg))


def average_precision(labels: np.ndarray, scores: np.ndarray) -> float:
    labels = np.asarray(labels).astype(np.int64).ravel()
    scores = np.asarray(scores, dtype=np.float64).ravel()
    n_pos = int((labels == 1).sum())
    if n_pos == 0:
        return float("nan")
    order = np.argsort(-scores, kind="mergesort")
    s = scores[order]
    y = labels[order] == 1
    tp = np.cumsum(y)
    fp = np.cumsum(~y)
    # thresholds = distinct score values: keep the last index of each tie group
    last = np.concatenate([np.flatnonzero(np.diff(s) != 0), [s.size - 1]])
    tp, fp = tp[last], fp[last]
    precision = tp / (tp + fp)
    recall = tp / n_pos
    prev_recall = np.concatenate([[0.0], recall[:-1]])
    return float(np.sum((recall - prev_recall) * precision))

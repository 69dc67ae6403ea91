"""Epoch-tail metrics of the trainer (`src/eoe/training/ad_trainer.py:452-455,516-522`): the reference calls
sklearn's `roc_curve` + `auc` (trapezoid) and `average_precision_score` on host copies of the epoch's labels and
scores.  Same place (host, once per epoch), written as the tie-aware rank statistic, which equals the trapezoidal
ROC area, and the step-wise precision-recall sum."""
import numpy as np


class ROC:
    """container mirroring `src/eoe/utils/logger.py:36-62` (only the score is kept: curves are plotting data)"""

    def __init__(self, auc: float, std: float = None, n: int = -1):
        self.auc, self.std, self.n = auc, std, n

    def get_score(self):
        return self.auc


class PRC:
    def __init__(self, avg_prec: float, std: float = None, n: int = -1):
        self.avg_prec, self.std, self.n = avg_prec, std, n

    def get_score(self):
        return self.avg_prec


def _avg_ranks(sorted_scores: np.ndarray) -> np.ndarray:
    n = sorted_scores.size
    bounds = np.flatnonzero(np.diff(sorted_scores) != 0) + 1
    starts = np.concatenate([[0], bounds])
    ends = np.concatenate([bounds, [n]])
    mid = 0.5 * (starts + ends - 1) + 1.0
    return np.repeat(mid, ends - starts)


def roc_auc(labels, scores) -> float:
    labels = np.asarray(labels).astype(np.int64).ravel()
    scores = np.asarray(scores, dtype=np.float64).ravel()
    pos = labels == 1
    n_pos = int(pos.sum())
    n_neg = labels.size - n_pos
    if n_pos == 0 or n_neg == 0:
        return float("nan")
    order = np.argsort(scores, kind="stable")
    ranks = np.empty(labels.size, dtype=np.float64)
    ranks[order] = _avg_ranks(scores[order])
    return float((ranks[pos].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg))


def average_precision(labels, scores) -> float:
    labels = np.asarray(labels).astype(np.int64).ravel()
    scores = np.asarray(scores, dtype=np.float64).ravel()
    n_pos = int((labels == 1).sum())
    if n_pos == 0:
        return float("nan")
    order = np.argsort(-scores, kind="stable")
    s, y = scores[order], labels[order] == 1
    tp, fp = np.cumsum(y), np.cumsum(~y)
    last = np.concatenate([np.flatnonzero(np.diff(s) != 0), [s.size - 1]])
    tp, fp = tp[last], fp[last]
    recall = tp / n_pos
    return float(np.sum(np.diff(np.concatenate([[0.0], recall])) * (tp / (tp + fp))))


def auc_ap_device(labels, scores):
    """(ROC AUC, average precision) of GPU-resident scores without the host round trip: `eoe_auc_ap` (exact integer pair counts,
    tie-aware; equals `roc_auc` / `average_precision` above up to fp64 rounding).  labels: int tensor [n] (1 = anomalous)."""
    import torch
    from ._lib import check, lib
    if not scores.is_cuda:
        raise RuntimeError("auc_ap_device needs GPU tensors (use roc_auc / average_precision on the host)")
    sc = scores.detach().reshape(-1).contiguous().float()
    la = labels.to(sc.device).reshape(-1).contiguous().to(torch.int64)
    n = sc.numel()
    out = torch.empty(2, dtype=torch.float64, device=sc.device)
    scratch = torch.empty(((n + 255) // 256) * 24, dtype=torch.uint8, device=sc.device)
    check(lib.eoe_auc_ap(sc.data_ptr(), la.data_ptr(), 1, out.data_ptr(), scratch.data_ptr(), n, torch.cuda.current_stream().cuda_stream),
          "eoe_auc_ap")
    auc, ap = out.cpu().tolist()
    return auc, ap

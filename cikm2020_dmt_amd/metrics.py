"""Streaming evaluation metrics of run_dnn.py:221-241: tf.metrics.auc (ROC, 200 thresholds, trapezoidal),
precision / recall at 0.5.  The per-batch confusion histogram is accumulated on the GPU (dmt_auc_hist);
the 200-bin suffix sums and the trapezoid are host arithmetic on 402 integers."""
from __future__ import annotations

import numpy as np
import torch

from . import _lib as L
from . import ops


def auc_from_hist(hist: np.ndarray, n_thr: int = 200) -> float:
    """hist[(label)*(n_thr+1) + k] = #examples with exactly k thresholds below the prediction."""
    h = np.asarray(hist, dtype=np.float64).reshape(2, n_thr + 1)
    # tp[i] = #positives with pred > t_i  = sum_{k > i} h[1, k]
    tp = np.array([h[1, i + 1:].sum() for i in range(n_thr)])
    fp = np.array([h[0, i + 1:].sum() for i in range(n_thr)])
    fn = h[1].sum() - tp
    tn = h[0].sum() - fp
    eps = 1.0e-6
    rec = (tp + eps) / (tp + fn + eps)
    fpr = fp / (fp + tn + eps)
    return float(((fpr[:-1] - fpr[1:]) * (rec[:-1] + rec[1:]) / 2.0).sum())


class StreamingAUC:
    def __init__(self, device, n_thr: int = 200):
        self.n_thr = n_thr
        self.hist = torch.zeros(2 * (n_thr + 1), dtype=torch.int64, device=device)

    def update(self, pred: torch.Tensor, label: torch.Tensor):
        pred = pred.reshape(-1).float().contiguous()
        label = label.reshape(-1).float().contiguous()
        L.call("dmt_auc_hist", pred.numel(), ops.p(pred), ops.p(label), self.n_thr, ops.p(self.hist), ops.stream_ptr())

    def result(self) -> float:
        return auc_from_hist(self.hist.cpu().numpy(), self.n_thr)

    def reset(self):
        self.hist.zero_()


class StreamingPrecisionRecall:
    """tf.metrics.precision / tf.metrics.recall as run_dnn.py:221-227, 230-238 use them: prediction = sigmoid score > 0.5
    (run_dnn.py:190, 195), label cast to bool; running true-positive / false-positive / false-negative totals, 0 when a
    denominator is 0 (tf.metrics' safe division).  The counts are accumulated on the GPU (dmt_confusion_counts)."""

    def __init__(self, device, threshold: float = 0.5):
        self.threshold = float(threshold)
        self.counts = torch.zeros(4, dtype=torch.int64, device=device)     # tp, fp, fn, tn

    def update(self, score: torch.Tensor, label: torch.Tensor):
        score = score.reshape(-1).float().contiguous()
        label = label.reshape(-1).float().contiguous()
        L.call("dmt_confusion_counts", score.numel(), ops.p(score), ops.p(label), self.threshold, ops.p(self.counts), ops.stream_ptr())

    def result(self):
        tp, fp, fn, _tn = (float(v) for v in self.counts.cpu().numpy())
        precision = tp / (tp + fp) if tp + fp > 0 else 0.0
        recall = tp / (tp + fn) if tp + fn > 0 else 0.0
        return precision, recall

    def reset(self):
        self.counts.zero_()


def precision_recall_reference(scores, labels, threshold: float = 0.5):
    """Host statement of the same rule (numpy): what tf.metrics.precision / recall return after streaming these batches."""
    s = np.concatenate([np.asarray(x, dtype=np.float32).reshape(-1) for x in scores])
    y = np.concatenate([np.asarray(x, dtype=np.float32).reshape(-1) for x in labels]) != 0
    p = s > np.float32(threshold)
    tp, fp, fn = float((p & y).sum()), float((p & ~y).sum()), float((~p & y).sum())
    return (tp / (tp + fp) if tp + fp > 0 else 0.0), (tp / (tp + fn) if tp + fn > 0 else 0.0)

"""Offline ranking metrics of the reference's evaluation (DMT_code/metrics/metrics.py:10-277; called from run_dnn.py:497-519 and
:780-800): per-session precision@N and MRR@N, and per-group AUC, for the click (label >= 2) and order (label >= 5) actions.

Vectorised numpy restatement (one lexsort + segment arithmetic instead of pandas groupby + a multiprocessing fan-out); the golden
vectors in tests/golden/offline_metrics_golden.json were produced by the reference module itself (tests/golden/make_metrics_golden.py).

  get_offline_metrics(header_schema, headers, scores) -> ({CLICK: (P@N[7], MRR@N[7]), ORDER: (...)}, at_list)
  get_offline_metrics_auc(header_schema, headers, scores, group_method='uuid'|'sid') -> {CLICK: [auc], ORDER: [auc]}

Reference rules kept: rows of a session are ranked by score descending, ties by label ascending (metrics.py:84); P@N divides by
min(N, session size); MRR@N is 1/rank of the first hit inside the top N; both are averaged over ALL sessions.  Group AUC skips
groups of one row, gives 1 to a group whose labels are all equal (the `except: return 1` around roc_auc_score, :66-74 -- the
scikit-learn the reference targeted raised there; 1.7 returns NaN with a warning instead) and averages over the remaining groups.
"""
from __future__ import annotations

from typing import Dict, Sequence, Tuple

import numpy as np

CLICK = 2
ORDER = 5
at_list = [2, 4, 6, 8, 10, 12, 14]


def _columns(header_schema: Sequence[str], headers, names: Sequence[str]):
    idx = [list(header_schema).index(n) for n in names]
    cols = [[] for _ in names]
    for h in headers:
        s = h.decode() if isinstance(h, (bytes, bytearray)) else str(h)
        parts = s.strip().split("\t")
        for c, i in zip(cols, idx):
            c.append(parts[i])
    return cols


def _groups(keys):
    """Stable group ids in order of first appearance of the SORTED key (pandas groupby sorts the keys; only sizes / means matter)."""
    uniq, inv = np.unique(np.asarray(keys, dtype=object).astype(str), return_inverse=True)
    return inv.astype(np.int64), len(uniq)


def _rank_within_groups(gid, score, label):
    """Row order (group, score desc, label asc) + each row's 0-based rank inside its group and the group sizes."""
    order = np.lexsort((label, -score, gid))
    g = gid[order]
    start = np.r_[0, np.flatnonzero(g[1:] != g[:-1]) + 1]
    sizes = np.diff(np.r_[start, len(g)])
    rank = np.arange(len(g)) - np.repeat(start, sizes)
    return order, g, rank, sizes


def get_offline_metrics(header_schema, headers, scores) -> Tuple[Dict[int, Tuple[np.ndarray, np.ndarray]], list]:
    (labels, sids) = _columns(header_schema, headers, ["label", "sid"])
    label = np.asarray(labels, dtype=np.int64)
    score = np.asarray(scores, dtype=np.float64).reshape(-1)
    gid, ng = _groups(sids)
    order, g, rank, sizes = _rank_within_groups(gid, score, label)
    lab = label[order]
    out = {}
    for action in (CLICK, ORDER):
        hit = (lab >= action)
        pre = np.zeros(len(at_list))
        mrr = np.zeros(len(at_list))
        # rank of the first hit of every group (inf if none)
        first = np.full(ng, np.inf)
        hr = np.where(hit, rank, np.inf)
        np.minimum.at(first, g, hr)
        for i, N in enumerate(at_list):
            top = rank < N
            cnt = np.bincount(g[top], weights=hit[top].astype(np.float64), minlength=ng)
            pre[i] = float(np.sum(cnt / np.minimum(sizes, N))) / ng
            mrr[i] = float(np.sum(np.where(first < N, 1.0 / (first + 1.0), 0.0))) / ng
        out[action] = (pre, mrr)
    return out, list(at_list)


def _auc(pos: np.ndarray, score: np.ndarray) -> float:
    """roc_auc_score for one group (ties count one half); all labels equal -> 1 (the reference's exception path)."""
    n1 = int(pos.sum())
    n0 = len(pos) - n1
    if n1 == 0 or n0 == 0:
        return 1.0
    order = np.argsort(score, kind="mergesort")
    s = score[order]
    # average ranks for ties
    rnk = np.empty(len(s), dtype=np.float64)
    i = 0
    while i < len(s):
        j = i
        while j + 1 < len(s) and s[j + 1] == s[i]:
            j += 1
        rnk[i:j + 1] = 0.5 * (i + j) + 1.0
        i = j + 1
    r = np.empty_like(rnk)
    r[order] = rnk
    return float((r[pos].sum() - n1 * (n1 + 1) / 2.0) / (n1 * n0))


def get_offline_metrics_auc(header_schema, headers, scores, group_method: str = "uuid") -> Dict[int, np.ndarray]:
    (labels, keys) = _columns(header_schema, headers, ["label", group_method])
    label = np.asarray(labels, dtype=np.int64)
    score = np.asarray(scores, dtype=np.float64).reshape(-1)
    gid, ng = _groups(keys)
    order = np.argsort(gid, kind="mergesort")
    g = gid[order]
    start = np.r_[0, np.flatnonzero(g[1:] != g[:-1]) + 1]
    end = np.r_[start[1:], len(g)]
    sums = {CLICK: 0.0, ORDER: 0.0}
    valid = 0
    for a, b in zip(start, end):
        if b - a == 1:
            continue
        rows = order[a:b]
        valid += 1
        for action in (CLICK, ORDER):
            sums[action] += _auc(label[rows] >= action, score[rows])
    return {CLICK: np.array([sums[CLICK] / valid]), ORDER: np.array([sums[ORDER] / valid])}

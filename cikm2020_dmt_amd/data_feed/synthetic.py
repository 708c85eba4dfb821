"""Synthetic batches of the jd_recsys schema at the post-vocabulary-lookup boundary.

Per example: one id per item field; clk / ord / cart behaviour sequences over (sku, ts, c2, c3, brand, shop);
`near_expo_seq_c2/c3` for the bias tower; 615 dense features; one-hot 5-class `mask`.
Index law: idx = 1 + Zipf(1.05) mod (vocab-1)  (or uniform);  Wts = 1.0;  dense ~ U(-0.99, 0.99) with 40 % zeros;
class probabilities from jd_recsys_demo/stat/stat/part-00000 (label counts).  Seed 20200101.
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np

from ..sparse import SparseTensorValue

CLASS_PROBS = np.array([0.93216, 0.00882, 0.05687, 0.00150, 0.00065])
DEFAULT_SEED = 20200101


def _draw_ids(rng, shape, vocab, law):
    if vocab <= 1:
        return np.zeros(shape, dtype=np.int64)
    if law == "uniform":
        return rng.integers(1, vocab, size=shape, dtype=np.int64)
    z = rng.zipf(1.05, size=shape).astype(np.uint64)
    return (1 + (z % np.uint64(vocab - 1))).astype(np.int64)


def make_batch(spec: dict, batch: int, seed: int = DEFAULT_SEED, lengths: str = "full", law: str = "zipf",
               seq_lens: Dict[str, int] = None, weights: str = "ones") -> Tuple[dict, np.ndarray, np.ndarray]:
    """Returns (inputs, mask[B,5] float32, label[B] float32).

    lengths: 'full' (every sequence at its maximum) or 'ragged' (len ~ U{1..L}).
    seq_lens: maximum length per sequence feature-name suffix; default from the feature names (_50 / _10).
    weights: 'ones' (as the demo data) or 'random' (U[0.5, 2.0], exercises the weighted-mean path).
    """
    rng = np.random.default_rng(seed)
    inputs: Dict[str, object] = {}
    vocab_of = {f: r for (_n, r, _d, f, _s) in spec["embedding_list"]}
    for (_n, r, _d, f, _s) in spec["embedding_list_bias"]:
        vocab_of.setdefault(f, r)

    def put(feat, idx_dense, lens):
        sp = SparseTensorValue.from_padded(idx_dense, lens)
        inputs[feat] = sp
        if weights == "random":
            w = rng.uniform(0.5, 2.0, size=len(sp.values)).astype(np.float32)
        else:
            w = np.ones(len(sp.values), dtype=np.float32)
        inputs[feat + "Wts"] = SparseTensorValue(sp.indices, w, sp.dense_shape)

    # item-side features: exactly one id per example
    item_feats = [f for (_n, _r, _d, f, s) in spec["embedding_list"] if s == "i"]
    for f in item_feats:
        put(f, _draw_ids(rng, (batch, 1), vocab_of[f], law), np.ones(batch, dtype=np.int64))

    # behaviour sequences: all fields of one sequence share the per-example length
    groups = [[p[0] for p in grp] for grp in spec["attention_embed_pairs"]]
    for gi, grp in enumerate(groups):
        feats = list(grp)
        if spec["attention_embed_seq_ts"]:
            feats.append(spec["attention_embed_seq_ts"][gi])
        L = None
        if seq_lens is not None:
            L = seq_lens.get(grp[0])
        if L is None:
            L = int(grp[0].rsplit("_", 1)[1])
        lens = np.full(batch, L, dtype=np.int64) if lengths == "full" else rng.integers(1, L + 1, size=batch)
        for f in feats:
            put(f, _draw_ids(rng, (batch, L), vocab_of[f], law), lens)

    # bias-tower features that are not already item features
    for (_n, r, _d, f, s) in spec["embedding_list_bias"]:
        if f in inputs:
            continue
        L = 6
        lens = np.full(batch, L, dtype=np.int64) if lengths == "full" else rng.integers(1, L + 1, size=batch)
        put(f, _draw_ids(rng, (batch, L), r, law), lens)

    dense = rng.uniform(-0.99, 0.99, size=(batch, spec["feature_dimension"])).astype(np.float32)
    dense[rng.random(dense.shape) < 0.4] = 0.0
    inputs["features"] = dense

    cls = rng.choice(5, size=batch, p=CLASS_PROBS / CLASS_PROBS.sum())
    mask = np.zeros((batch, 5), dtype=np.float32)
    mask[np.arange(batch), cls] = 1.0
    label = (cls > 0).astype(np.float32)
    return inputs, mask, label


def slice_batch(spec: dict, inputs: dict, mask: np.ndarray, a: int, b: int):
    """Examples [a, b) of a batch made by make_batch (the towers of run_dnn.py:148-207 each take such a slice of the input queue)."""
    out = {}
    for k, v in inputs.items():
        if isinstance(v, SparseTensorValue):
            idx = np.asarray(v.indices)
            sel = (idx[:, 0] >= a) & (idx[:, 0] < b)
            ind = idx[sel].copy()
            ind[:, 0] -= a
            out[k] = SparseTensorValue(ind, np.asarray(v.values)[sel], (b - a,) + tuple(v.dense_shape[1:]))
        else:
            out[k] = np.asarray(v)[a:b]
    return out, mask[a:b]


def write_records_file(job):
    """(dims, sku_rows, batch, seed, law, path) -> number of bytes written.  One file of `batch` synthetic tf.Example records of the
    model's TFRecord schema (data_feed/tfrecord_mask.py:23-84: every id feature a bytes list + its ...Wts float list, `features`,
    `mask`, `label`), the same id law as make_batch.  A top-level function: bench.py runs it in spawned worker processes (the Python
    protobuf encoder does ~1 ms per 10 KB record)."""
    import os
    from .. import spec as S
    from . import tfrecord
    dims, sku_rows, batch, seed, law, path = job
    sp = S.e64_spec() if dims == "e64" else S.default_spec()
    if sku_rows:
        sp = S.scaled_spec(sp, {"Sku": sku_rows})
    emb = list(sp["embedding_list"]) + list(sp["embedding_list_bias"])
    feats = list(dict.fromkeys(e[3] for e in emb))
    inputs, mask, label = make_batch(sp, batch, seed=seed, lengths="full", law=law)
    rows = {f: inputs[f].rows() for f in feats}
    recs = []
    for b in range(batch):
        ex = {"features": inputs["features"][b].astype(np.float32), "mask": mask[b].astype(np.float32), "label": np.array([label[b]], np.float32)}
        for f in feats:
            ex[f] = [("%d" % int(i)).encode() for i in rows[f][b]]
            ex[f + "Wts"] = np.ones(len(rows[f][b]), np.float32)
        recs.append(tfrecord.encode_example(ex))
    tfrecord.write_records(path, recs)
    return os.path.getsize(path)

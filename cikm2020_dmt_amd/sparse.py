"""SparseTensorValue: the container the reference's input pipeline hands to the model.

Mirrors what `tf.parse_example` + `LookupTables.transform_id2index` produce for every id feature
(/root/reference/DMT_code/data_feed/tfrecord_mask.py:23-84, data_feed/index_tables.py:37-45):
row-major `indices [nnz,2] int64`, `values [nnz]` (int64 ids after vocabulary lookup, float32 for
`<name>Wts`), `dense_shape (B, max_len_in_batch)`.
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np


class SparseTensorValue:
    __slots__ = ("indices", "values", "dense_shape")

    def __init__(self, indices, values, dense_shape):
        self.indices = np.ascontiguousarray(np.asarray(indices, dtype=np.int64).reshape(-1, 2))
        self.values = np.ascontiguousarray(np.asarray(values))
        self.dense_shape = (int(dense_shape[0]), int(dense_shape[1]))

    @staticmethod
    def from_rows(rows: Sequence[Sequence], dtype) -> "SparseTensorValue":
        lens = np.fromiter((len(r) for r in rows), dtype=np.int64, count=len(rows))
        T = int(lens.max()) if len(rows) else 0
        b = np.repeat(np.arange(len(rows), dtype=np.int64), lens)
        t = np.concatenate([np.arange(n, dtype=np.int64) for n in lens]) if len(rows) else np.zeros(0, np.int64)
        vals = np.concatenate([np.asarray(r, dtype=dtype) for r in rows]) if len(rows) else np.zeros(0, dtype)
        return SparseTensorValue(np.stack([b, t], 1), vals, (len(rows), T))

    @staticmethod
    def from_padded(dense: np.ndarray, lens: np.ndarray) -> "SparseTensorValue":
        """Left-aligned rows: entry (b,t) exists iff t < lens[b]."""
        B, T = dense.shape
        valid = np.arange(T)[None, :] < np.asarray(lens)[:, None]
        b, t = np.nonzero(valid)
        Tm = int(np.max(lens)) if B else 0
        return SparseTensorValue(np.stack([b, t], 1), dense[valid], (B, Tm))

    def lengths(self) -> np.ndarray:
        """Number of entries per row (what reduce_sum(to_dense(ones)) gives; mmoe_transformer.py:137-142)."""
        return np.bincount(self.indices[:, 0], minlength=self.dense_shape[0]).astype(np.int32)

    def to_padded(self, T: int = None, fill=0):
        """(dense [B,T], lens [B]).  Requires left-aligned rows (column index == rank within the row),
        which is what VarLenFeature batching produces."""
        B = self.dense_shape[0]
        T = self.dense_shape[1] if T is None else T
        out = np.full((B, T), fill, dtype=self.values.dtype)
        if len(self.values):
            out[self.indices[:, 0], self.indices[:, 1]] = self.values
        return out, self.lengths()

    def rows(self) -> List[np.ndarray]:
        lens = self.lengths()
        off = np.concatenate([[0], np.cumsum(lens)])
        return [self.values[off[i]:off[i + 1]] for i in range(self.dense_shape[0])]

"""Loading / renumbering helpers shared by tests/golden/make_golden.py and the tests that consume the fixtures."""
import os

import numpy as np

from cikm2020_dmt_amd.sparse import SparseTensorValue

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_demo():
    z = np.load(os.path.join(GOLDEN, "demo474.npz"))
    return {k: z[k] for k in z.files}


def build_inputs(demo, spec):
    """{'features', f, f+'Wts'} with SparseTensorValue columns for every feature of the spec (all 474 examples)."""
    inputs = {"features": demo["features"]}
    feats = [f for (_n, _r, _d, f, _s) in spec["embedding_list"]] + [f for (_n, _r, _d, f, _s) in spec["embedding_list_bias"]]
    for f in dict.fromkeys(feats):
        lens = demo["l_" + f].astype(np.int64)
        vals = demo["v_" + f].astype(np.int64)
        b = np.repeat(np.arange(len(lens)), lens)
        t = np.concatenate([np.arange(n) for n in lens])
        shape = (len(lens), int(lens.max()))
        inputs[f] = SparseTensorValue(np.stack([b, t], 1), vals, shape)
        w = demo["w_" + f] if ("w_" + f) in demo else np.ones(len(vals), np.float32)
        inputs[f + "Wts"] = SparseTensorValue(np.stack([b, t], 1), w.astype(np.float32), shape)
    return inputs


def compact_inputs(inputs, spec):
    """Renumber the rows each vocabulary touches to 0..|S|-1 with S = sorted({idx} U {idx-1} U {0}); returns
    (new inputs, {'rows': {vocab: |S|}, 'maps': {vocab: S}})."""
    vocab_of = {}
    for (n, _r, _d, f, _s) in list(spec["embedding_list"]) + list(spec["embedding_list_bias"]):
        vocab_of.setdefault(f, n)
    used = {}
    for f, n in vocab_of.items():
        v = np.asarray(inputs[f].values, dtype=np.int64)
        used.setdefault(n, []).append(v)
    maps, rows = {}, {}
    for n, lst in used.items():
        v = np.unique(np.concatenate(lst))
        s = np.unique(np.concatenate([v, np.maximum(v - 1, 0), [0]]))
        maps[n], rows[n] = s, int(len(s))
    out = dict(inputs)
    for f, n in vocab_of.items():
        sp = inputs[f]
        out[f] = SparseTensorValue(sp.indices, np.searchsorted(maps[n], np.asarray(sp.values, dtype=np.int64)), sp.dense_shape)
    return out, dict(rows=rows, maps=maps)


def batch_slice(inputs, mask, ids, spec):
    """Sub-batch of examples `ids` (in that order); dense_shape[1] = longest row of the sub-batch."""
    ids = np.asarray(ids)
    out = {"features": inputs["features"][ids]}
    for k, sp in inputs.items():
        if k == "features":
            continue
        rows = sp.rows()
        sel = [rows[i] for i in ids]
        out[k] = SparseTensorValue.from_rows(sel, sp.values.dtype)
    return out, mask[ids]


def train_schedule(n_examples, batch, steps):
    """Deterministic stand-in for the reference's shuffle: consecutive windows over the examples, wrapping around."""
    for s in range(steps):
        yield (np.arange(batch) + s * batch) % n_examples

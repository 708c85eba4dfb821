"""bf16 engine against the fp32 engine of the same library at E64 dims (same parameters, same batch, same dropout counters):
per-tensor relative gradient error; an outlier tensor would point at a bf16-only kernel path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cikm2020_dmt_amd import spec as S
from cikm2020_dmt_amd.data_feed.synthetic import make_batch
from cikm2020_dmt_amd.train import Trainer
from tests.util import SMALL_ROWS, sparse_to_dense_tables
sp = S.scaled_spec(S.e64_spec(), SMALL_ROWS)
for (B, drop, lengths) in [(512, False, "ragged"), (512, True, "full"), (37, True, "ragged")]:
    inputs, mask, label = make_batch(sp, B, seed=9, lengths=lengths, weights="random")
    res = []
    for dt in (torch.float32, torch.bfloat16):
        tr = Trainer(sp, device="cuda:0", compute_dtype=dt, seed=4, dropout=drop, dropout_seed=77)
        loss = float(tr.forward_backward(tr.make_batch(inputs, mask, label)))
        g = dict(tr.store.grad_dict()); g.update(sparse_to_dense_tables(tr.store, tr.engine.sparse))
        res.append((loss, g))
    errs = []
    gscale = max(float(np.abs(v).max()) for v in res[0][1].values())
    for name in res[0][1]:
        a, b = res[0][1][name], res[1][1][name]
        # (tensors whose true gradient is ~0, e.g. the key bias the softmax is invariant to, are measured against the global scale)
        errs.append((float(np.linalg.norm(a - b) / max(np.linalg.norm(a), 3e-3 * gscale * np.sqrt(a.size))),
                     float(np.abs(a - b).max() / max(np.abs(a).max(), 3e-2 * gscale)), name))
    errs.sort(reverse=True)
    print("B=%d dropout=%s %s: loss fp32 %.5f bf16 %.5f; worst tensors (rel L2, rel max):" % (B, drop, lengths, res[0][0], res[1][0]))
    for e in errs[:4]:
        print("   %.3g %.3g %s" % e)

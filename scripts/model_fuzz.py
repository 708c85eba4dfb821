"""fp32 forward/backward of the small-vocabulary model against the oracle for several batch sizes and seeds, element by element
(chunk borders of the sorted embedding-gradient reduce move with the batch size)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import dmt_oracle as O
from oracle import dmt_oracle_torch as OT
from cikm2020_dmt_amd.data_feed.synthetic import make_batch
from cikm2020_dmt_amd.train import Trainer
from tests.util import small_specs, sparse_to_dense_tables
so, sp = small_specs()
worst = (0.0, None)
for B in (1, 7, 64, 300):
    for seed in (1, 2, 3):
        P = O.init_params(so, seed=seed)
        inputs, mask, label = make_batch(sp, B, seed=100 * seed + B, lengths="ragged", weights="random")
        tr = Trainer(sp, device="cuda:0", compute_dtype=torch.float32, init=False, dropout=False)
        tr.store.load_state(P)
        loss_ref, (c_ref, o_ref, yb_ref), G = OT.loss_and_grads(P, inputs, mask, so)
        loss = tr.forward_backward(tr.make_batch(inputs, mask, label))
        (c, o), yb = tr.last["out"]
        e_fwd = max(np.abs(c.detach().cpu().numpy() - c_ref).max(), np.abs(o.detach().cpu().numpy() - o_ref).max())
        got = dict(tr.store.grad_dict()); got.update(sparse_to_dense_tables(tr.store, tr.engine.sparse))
        gscale = max(np.abs(G[n]).max() for n in got)
        e_g, where = 0.0, None
        for name, g in got.items():
            e = np.abs(g - G[name]).max() / max(np.abs(G[name]).max(), 1e-3 * gscale)
            if e > e_g:
                e_g, where = e, name
        print("B=%3d seed=%d  logits %.2e  loss rel %.2e  worst grad element %.2e (%s)" % (B, seed, e_fwd, abs(float(loss) - loss_ref) / abs(loss_ref), e_g, where))
        if e_g > worst[0]:
            worst = (e_g, (B, seed, where))
print("worst", worst)

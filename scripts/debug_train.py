import numpy as np, torch, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import dmt_oracle as O, dmt_oracle_torch as OT
from cikm2020_dmt_amd.data_feed.synthetic import make_batch
from cikm2020_dmt_amd.train import Trainer
from tests.util import small_specs, sparse_to_dense_tables
so, sp = small_specs()
P = O.init_params(so, seed=11)
tr = Trainer(sp, device='cuda', compute_dtype=torch.float32, init=False)
tr.store.load_state(P)
Pn = {k: v.copy() for k, v in P.items()}
adam = O.TFAdam(lr=1e-3)
name = 'mmoe_layers/expert-0/expert-layer-0/weights'
for i in range(3):
    inp, m, _ = make_batch(sp, 16, seed=100+i, lengths='ragged', weights='random')
    _l, _lg, G = OT.loss_and_grads(Pn, inp, m, so)
    b = tr.make_batch(inp, m)
    tr.forward_backward(b)
    g = tr.store.grad_dict()[name]
    ref = G[name]
    d = np.abs(g-ref)
    print('step', i, 'grad maxabs ref', np.abs(ref).max(), 'max abs diff', d.max(), 'at', np.unravel_index(d.argmax(), d.shape))
    small = (np.abs(ref) < 1e-6) & (np.abs(ref) > 0)
    print('   #small ref elems', small.sum(), ' #exact zero ref', (ref==0).sum(), ' #exact zero mine', (g==0).sum(), ' mine nonzero where ref zero', ((ref==0)&(g!=0)).sum(), np.abs(g[(ref==0)]).max() if (ref==0).any() else None)
    adam.apply(Pn, G)
    tr.opt.step(tr.engine.sparse)
    got = tr.store.state_dict()[name]
    dd = np.abs(got - Pn[name])
    print('   param diff max', dd.max(), ' count>1e-5', (dd>1e-5).sum(), 'cols with bad', np.unique(np.nonzero(dd>1e-5)[1])[:20], 'rows', np.unique(np.nonzero(dd>1e-5)[0])[:20])
    if (dd>1e-5).any():
        r,c = np.nonzero(dd>1e-5); r,c=r[0],c[0]
        print('   example elem', r, c, 'ref grad', ref[r,c], 'my grad', g[r,c], 'ref p', Pn[name][r,c], 'my p', got[r,c], 'orig', P[name][r,c])

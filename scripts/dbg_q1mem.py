import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cikm2020_dmt_amd import spec as S
from cikm2020_dmt_amd.data_feed.synthetic import make_batch
from cikm2020_dmt_amd.train import Trainer
sp = S.scaled_spec(S.e64_spec(), {"Sku": 2000, "Brand": 300, "Shopid": 300, "Cid3": 120})
inputs, mask, _ = make_batch(sp, 40, seed=6, lengths="ragged", weights="random")
out = []
for raw in (True, False):
    tr = Trainer(sp, device="cuda", compute_dtype=torch.bfloat16, seed=9, dropout=False)
    tr.engine.use_q1mem = raw
    eng = tr.engine
    grads = {}
    orig = eng.mha_cross
    def spy(q_in, mem, q_lens, k_lens, blk, stream=3, orig=orig, grads=grads):
        mem.register_hook(lambda g, blk=blk: grads.__setitem__(blk + "dmem", g.detach().float().clone()))
        q_in.register_hook(lambda g, blk=blk: grads.__setitem__(blk + "dy", g.detach().float().clone()))
        s = orig(q_in, mem, q_lens, k_lens, blk, stream)
        s.register_hook(lambda g, blk=blk: grads.__setitem__(blk + "ds", g.detach().float().clone()))
        grads[blk + "s"] = s.detach().float().clone()
        grads[blk + "mem"] = mem.detach().float().clone()
        grads[blk + "klen"] = k_lens.clone()
        return s
    eng.mha_cross = spy
    loss = tr.forward_backward(tr.make_batch(inputs, mask))
    out.append(grads)
a, b = out
for k in sorted(b):
    if k.endswith("klen"): continue
    d = (a[k] - b[k]).abs().max().item(); m = b[k].abs().max().item()
    print("%-70s maxdiff %.4g of %.4g  (%.3f)" % (k[-70:], d, m, d / (m + 1e-12)))
k = [x for x in b if x.endswith("dmem")][0]
da, db = a[k], b[k]
bad = ((da - db).abs() > 0.1 * db.abs().max()).nonzero()
print("bad dmem entries", bad.shape, bad[:10].tolist(), "klen", b[k[:-4] + "klen"].tolist()[:12])

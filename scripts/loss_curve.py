"""Loss trajectory of the bench workload (same seeds as bench.py): python scripts/loss_curve.py [steps] [--no-dropout]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cikm2020_dmt_amd import spec as S
from cikm2020_dmt_amd.data_feed.synthetic import make_batch
from cikm2020_dmt_amd.train import Trainer
steps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 25
drop = "--no-dropout" not in sys.argv
sp = S.e64_spec()
tr = Trainer(sp, device="cuda", compute_dtype=torch.bfloat16, seed=1234, dropout=drop)
bs = []
for i in range(4):
    inputs, mask, label = make_batch(sp, 4096, seed=20200101 + i, lengths="full")
    bs.append(tr.make_batch(inputs, mask, label))
out = []
for i in range(steps):
    b = bs[i % 4]; b._prep = None
    out.append(float(tr.train_step(b)))
print(" ".join("%.4f" % x for x in out))

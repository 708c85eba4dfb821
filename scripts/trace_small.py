"""List the small aten ops (fill / copy / add / cat ...) of one train step with their shapes: what is left outside the C-ABI kernels."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from cikm2020_dmt_amd import spec as S
from cikm2020_dmt_amd.data_feed.synthetic import make_batch
from cikm2020_dmt_amd.train import Trainer

sp = S.e64_spec()
dev = torch.device("cuda", 0)
tr = Trainer(sp, device=dev, compute_dtype=torch.bfloat16, seed=1, dropout=True)
inputs, mask, label = make_batch(sp, 4096, seed=3, lengths="full", law="zipf")
b = tr.make_batch(inputs, mask, label)
for _ in range(3):
    b._prep = None
    tr.train_step(b)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    b._prep = None
    tr.train_step(b)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True, group_by_stack_n=12):
    if not e.key.startswith("aten::") or e.self_device_time_total <= 0:
        continue
    st = [f for f in (e.stack or []) if "cikm2020_dmt_amd" in f]
    rows.append((e.self_device_time_total, e.key, str(e.input_shapes)[:70], st[0].split("cikm2020_dmt_amd/")[-1][:60] if st else "?", e.count))
for t, k, shp, st, n in sorted(rows, reverse=True)[:70]:
    print("%-16s %-72s %-62s n=%3d %8.1f us" % (k, shp, st, n, t))

"""Which torch (aten) kernels does one train step launch, and from which source line?  (GPU box)"""
import sys, collections, torch
sys.path.insert(0, ".")
from torch.profiler import profile, ProfilerActivity
import bench
from cikm2020_dmt_amd.train import Trainer
from cikm2020_dmt_amd import spec as S
from cikm2020_dmt_amd.data_feed.synthetic import make_batch
sp = S.e64_spec()
dev = torch.device("cuda:0")
tr = Trainer(sp, device=dev, compute_dtype=torch.bfloat16, dropout=True)
inputs, mask, label = make_batch(sp, 4096, seed=1, lengths="full", weights="ones")
b = tr.make_batch(inputs, mask, label)
for _ in range(3): tr.train_step(b)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.train_step(b)
    torch.cuda.synchronize()
import traceback
rows = prof.key_averages(group_by_stack_n=12)
sel = [r for r in rows if r.key in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::add", "aten::add_", "aten::mul", "aten::clone", "aten::_to_copy", "aten::zeros", "aten::cat", "aten::sum", "aten::index_select", "aten::slice_backward")]
sel.sort(key=lambda r: -(r.device_time_total if hasattr(r, "device_time_total") else r.cuda_time_total))
for r in sel[:40]:
    st = [f for f in r.stack if ("cikm2020" in f or "bench" in f)]
    t = r.device_time_total if hasattr(r, "device_time_total") else r.cuda_time_total
    print("%3d x %-14s %7.1f us | %s" % (r.count, r.key, t, " <- ".join(x.split("/")[-1][:60] for x in st[:3])))

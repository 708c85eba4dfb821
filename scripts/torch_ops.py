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
agg = collections.Counter(); tim = collections.Counter()
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CPU and ev.name.startswith("aten::") and ev.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::add", "aten::add_", "aten::mul", "aten::clone", "aten::contiguous", "aten::to", "aten::_to_copy", "aten::zeros", "aten::empty_like", "aten::sum", "aten::cat", "aten::index", "aten::select_backward", "aten::slice_backward"):
        st = [f for f in (ev.stack or []) if "cikm2020_dmt_amd" in f or "bench" in f]
        key = (ev.name, st[0].strip() if st else "(autograd engine)")
        agg[key] += 1; tim[key] += ev.device_time_total if hasattr(ev, "device_time_total") else ev.cuda_time_total
for k, n in sorted(agg.items(), key=lambda kv: -tim[kv[0]])[:60]:
    print("%4d x %-18s %8.1f us  %s" % (n, k[0], tim[k], k[1][:150]))

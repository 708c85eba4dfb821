"""When does each sequence lane's backward get going, WITHOUT a profiler slowing the host down?  Events recorded from gradient hooks:
t0 = dL/dz exists (compute stream), t_i = the decoder of sequence i has handed dL/dmemory_i to its encoder (on lane i).  ms since t0."""
import sys
import torch
sys.path.insert(0, ".")
from cikm2020_dmt_amd import spec as S
from cikm2020_dmt_amd.data_feed.synthetic import make_batch
from cikm2020_dmt_amd.train import Trainer

dev = torch.device("cuda:0")
sp = S.e64_spec()
tr = Trainer(sp, device=dev, compute_dtype=torch.bfloat16, seed=1, dropout=True)
bs = []
for i in range(4):
    inputs, mask, label = make_batch(sp, 4096, seed=i, lengths="full")
    bs.append(tr.make_batch(inputs, mask, label))
for i in range(8):
    bs[i % 4]._prep = None
    tr.train_step(bs[i % 4], prefetch=bs[(i + 1) % 4])
torch.cuda.synchronize()
eng = tr.engine
orig = eng.embedding_trans
marks = {}

def patched(batch):
    z = orig(batch)
    ev0 = torch.cuda.Event(enable_timing=True)
    z.register_hook(lambda g: (ev0.record(torch.cuda.current_stream(dev)), None)[1])
    marks["z"] = ev0
    for i in range(3):
        mem = eng.intermediates["memory_%d" % i]
        ev = torch.cuda.Event(enable_timing=True)
        mem.register_hook(lambda g, ev=ev: (ev.record(torch.cuda.current_stream(dev)), None)[1])
        marks[i] = ev
    return z

eng.embedding_trans = patched
for rep in range(3):
    for i in range(4):
        bs[i]._prep = None
        tr.train_step(bs[i], prefetch=bs[(i + 1) % 4])
    torch.cuda.synchronize()
    print("decoder backward of sequence 0 / 1 / 2 done at  %.2f  %.2f  %.2f ms after dL/dz" % tuple(marks["z"].elapsed_time(marks[i]) for i in range(3)))

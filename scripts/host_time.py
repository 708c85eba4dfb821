"""Host (enqueue) time per training step vs the GPU's: is the step launch-bound?  usage: python scripts/host_time.py [--profile]"""
import sys
import time
import torch
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
def step(i):
    b = bs[i % 4]; b._prep = None
    return tr.train_step(b)
for i in range(6):
    step(i)
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter(); hs = []
    for i in range(20):
        a = time.perf_counter(); step(i); hs.append(time.perf_counter() - a)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("host enqueue %.2f ms/step (min %.2f)  wall %.2f ms/step  drain after last enqueue %.2f ms" % ((t1 - t0) / 20 * 1e3, min(hs) * 1e3, (t2 - t0) / 20 * 1e3, (t2 - t1) * 1e3))
if "--profile" in sys.argv:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for i in range(10):
        step(i)
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(35)

"""Per-launch timing of the GEMM calls of one train step (HIP events), grouped by shape/layout."""
import sys, os, collections
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cikm2020_dmt_amd import ops, spec as S
from cikm2020_dmt_amd.data_feed.synthetic import make_batch
from cikm2020_dmt_amd.train import Trainer
dims = sys.argv[1] if len(sys.argv) > 1 else 'e64'
dt = torch.bfloat16 if (len(sys.argv) < 3 or sys.argv[2] == 'bf16') else torch.float32
sp = S.e64_spec() if dims == 'e64' else S.default_spec()
tr = Trainer(sp, device='cuda', compute_dtype=dt, seed=1, dropout=False)
inputs, mask, label = make_batch(sp, 4096, seed=1, lengths='full')
b = tr.make_batch(inputs, mask, label)
orig = ops.gemm
log = []
def gemm_logged(A, a_rs, a_cs, Bm, b_rs, b_cs, M, N, K, out, ldc, **kw):
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); orig(A, a_rs, a_cs, Bm, b_rs, b_cs, M, N, K, out, ldc, **kw); e1.record()
    am = 'k' if a_cs == 1 else ('m' if a_rs == 1 else 's'); bm = 'k' if b_rs == 1 else ('n' if b_cs == 1 else 's')
    log.append(((M, N, K, am, bm, kw.get('split_k', 1)), e0, e1))
ops.gemm = gemm_logged
for i in range(3):
    b._prep = None; tr.train_step(b)
log.clear()
torch.cuda.synchronize()
import time
t0 = time.perf_counter(); b._prep = None; tr.train_step(b); torch.cuda.synchronize(); t1 = time.perf_counter()
agg = collections.OrderedDict()
for key, e0, e1 in log:
    ms = e0.elapsed_time(e1)
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += ms
tot = 0
print("step wall %.2f ms" % ((t1 - t0) * 1e3))
for key, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    M, N, K, am, bm, sk = key
    fl = 2.0 * M * N * K * n
    print("M=%7d N=%5d K=%7d A:%s B:%s split=%3d  n=%2d  %8.3f ms  %7.1f TF/s" % (M, N, K, am, bm, sk, n, ms, fl / ms / 1e9))
    tot += ms
print("total gemm ms", tot)

"""Which GEMM shapes one train step launches (M, N, K, batch, operand layouts, epilogue), with counts."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cikm2020_dmt_amd import ops, spec as S
from cikm2020_dmt_amd.data_feed.synthetic import make_batch
from cikm2020_dmt_amd.train import Trainer
sp = S.e64_spec()
tr = Trainer(sp, device="cuda:0", compute_dtype=torch.bfloat16, seed=1, dropout=True)
inputs, mask, label = make_batch(sp, 4096, seed=3, lengths="full", law="zipf")
b = tr.make_batch(inputs, mask, label)
tr.train_step(b)
seen = collections.Counter()
orig = ops.gemm
def spy(A, a_rs, a_cs, Bm, b_rs, b_cs, M, N, K, out, ldc, **kw):
    key = (M, N, K, kw.get("batch", 1), "A%s" % ("k" if a_cs == 1 else "m"), "B%s" % ("k" if b_rs == 1 else "n"), "f32" if out.dtype == torch.float32 else "bf16",
           "split%d" % kw.get("split_k", 1), "bias" if kw.get("bias") is not None else "", "relu" if kw.get("act_ncols") else "", "resid" if kw.get("resid") is not None else "",
           "gate" if kw.get("gate") is not None else "")
    seen[key] += 1
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = orig(A, a_rs, a_cs, Bm, b_rs, b_cs, M, N, K, out, ldc, **kw)
    e1.record()
    evs.setdefault(key, []).append((e0, e1))
    return r
evs = {}
ops.gemm = spy
b._prep = None
tr.train_step(b)
torch.cuda.synchronize()
tot = 0.0
for k, n in sorted(seen.items(), key=lambda kv: -sum(a.elapsed_time(b) for a, b in evs[kv[0]])):
    t = sum(a.elapsed_time(b) for a, b in evs[k])
    tot += t
    fl = 2.0 * k[0] * k[1] * k[2] * k[3] * n
    print("%2d x %-90s %7.1f us total  %6.1f TF/s" % (n, str(k), t * 1e3, fl / (t * 1e-3) / 1e12 if t > 0 else 0))
print("all dmt_gemm launches: %.3f ms" % tot)

"""Time the index plane's sort (dmt_sort_pairs + dmt_segment_heads, csrc/dmt_sort.hip) on the bench's key law: n = 1220 entries x B
Zipf(1.05) global rows < 2^23.   python scripts/sort_bench.py [B]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cikm2020_dmt_amd import _lib as L
from cikm2020_dmt_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = 1220 * B
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
total_rows = 5_432_523
keys = (rng.zipf(1.05, n) % total_rows).astype(np.uint32)
k = torch.as_tensor(keys.view(np.int32)).to(dev)
ks, vs = torch.empty_like(k), torch.empty_like(k)
seg, uniq, nu = torch.empty_like(k), torch.empty_like(k), torch.empty((1,), dtype=torch.int32, device=dev)
end_bit = total_rows.bit_length()
need = C.c_uint64(0)
L.call("dmt_sort_pairs", ops.p(k), ops.p(ks), None, ops.p(vs), n, end_bit, None, C.byref(need), ops.stream_ptr())
ws = torch.empty((int(need.value),), dtype=torch.uint8, device=dev)
need2 = C.c_uint64(0)
L.call("dmt_segment_heads", ops.p(ks), n, total_rows, ops.p(seg), ops.p(uniq), ops.p(nu), None, C.byref(need2), ops.stream_ptr())
ws2 = torch.empty((int(need2.value),), dtype=torch.uint8, device=dev)


def run():
    have = C.c_uint64(ws.numel())
    L.call("dmt_sort_pairs", ops.p(k), ops.p(ks), None, ops.p(vs), n, end_bit, ops.p(ws), C.byref(have), ops.stream_ptr())
    have2 = C.c_uint64(ws2.numel())
    L.call("dmt_segment_heads", ops.p(ks), n, total_rows, ops.p(seg), ops.p(uniq), ops.p(nu), ops.p(ws2), C.byref(have2), ops.stream_ptr())


for _ in range(5):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    run()
e1.record()
torch.cuda.synchronize()
print("n = %d pairs, %d key bits: sort + segmentation %.1f us (%.2f G pairs/s); distinct rows %d" % (n, end_bit, e0.elapsed_time(e1) * 20.0, n / (e0.elapsed_time(e1) * 20e-6) / 1e9, int(nu.item())))
order = np.argsort(keys, kind="stable")
assert np.array_equal(ks.cpu().numpy().view(np.uint32), keys[order]) and np.array_equal(vs.cpu().numpy().view(np.uint32), order.astype(np.uint32))
print("matches numpy's stable argsort")

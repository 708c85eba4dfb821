"""Timing of the B-row GEMM shapes of a step in isolation (generic dmt_gemm), to see what the small launches cost and why."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cikm2020_dmt_amd import ops
dev = torch.device("cuda")
BF, F32 = torch.bfloat16, torch.float32
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
B = 4096
x = torch.randn(B, 4 * 320, device=dev).to(BF); q = torch.randn(B, 320, device=dev).to(BF)
for (Kd, N, batch, split) in [(320, 80, 4, 8), (320, 80, 4, 1), (320, 80, 1, 8), (320, 320, 1, 8), (320, 128, 4, 8), (320, 128, 1, 8), (384, 128, 1, 8), (320, 64, 4, 8), (256, 128, 4, 8), (1, 80, 4, 8)]:
    gw = torch.zeros(Kd, 4 * max(N, 80), device=dev)
    xx = torch.randn(B, 4 * max(Kd, 8), device=dev).to(BF)
    qq = torch.randn(B, 4 * N, device=dev).to(BF)
    us = t(lambda: ops.gemm(xx, 1, xx.stride(0), qq, qq.stride(0), 1, Kd, N, B, gw, gw.stride(0), split_k=split, accumulate=True, batch=batch, a_bs=Kd, b_bs=N, c_bs=N))
    print("wgrad  Kd=%4d N=%4d batch=%d split=%d : %6.1f us" % (Kd, N, batch, split, us))
w = torch.randn(4 * 320, 320, device=dev).to(BF)
for (N, K, batch, odt) in [(320, 80, 4, F32), (320, 80, 4, BF), (320, 80, 1, BF), (80, 320, 4, BF), (80, 328, 4, BF), (320, 320, 1, BF), (128, 320, 4, BF), (128, 128, 1, BF), (32, 128, 1, BF)]:
    a = torch.randn(B, 4 * K, device=dev).to(BF)
    wt = torch.randn(4 * N, K, device=dev).to(BF)
    out = torch.empty(B, 4 * N, dtype=odt, device=dev)
    us = t(lambda: ops.gemm(a, a.stride(0), 1, wt, 1, K, B, N, K, out, out.stride(0), batch=batch, a_bs=K, b_bs=N * K, c_bs=N))
    print("fwd    N=%4d K=%4d batch=%d out=%s : %6.1f us" % (N, K, batch, "f32" if odt == F32 else "bf16", us))
z = torch.empty(1, device=dev)
print("empty-ish launch (dmt_dropout n=1): %.1f us" % t(lambda: ops.L.call("dmt_dropout", 1, 64, ops.p(x), ops.p(x), 1, 0.5, ops.stream_ptr())))

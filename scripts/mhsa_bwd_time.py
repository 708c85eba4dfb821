"""Time dmt_mhsa_block_bwd against the two launches it replaces (dmt_attn_bwd + the dx GEMM) at bench size.
    python scripts/mhsa_bwd_time.py [T] [B]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cikm2020_dmt_amd import _lib
if os.environ.get("DMT_LIB_OVERRIDE"):          # an experimental build (make EXPERIMENTS=1) kept beside the shipped library
    _lib.LIB_PATH = os.path.abspath(os.environ["DMT_LIB_OVERRIDE"])
from cikm2020_dmt_amd import ops

dev = torch.device("cuda")
T = int(sys.argv[1]) if len(sys.argv) > 1 else 50
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
d, H = 320, 4
BF = torch.bfloat16
w = torch.randn(d, 3 * d, device=dev) * (1.0 / d) ** 0.5
wt = ops.Weight(w, w.to(BF), w.to(BF).t().contiguous())
imgb = torch.empty(ops.mhsa_bwd_image_bytes(), dtype=torch.uint8, device=dev)
ops.mhsa_bwd_image_build(w, imgb)
qkv = torch.randn(B, T, 3 * d, device=dev).to(BF)
ds = torch.randn(B, T, d, device=dev).to(BF)
lens = torch.full((B,), T, dtype=torch.int32, device=dev)
ref = torch.empty_like(qkv)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for keep in (1.0, 0.9):
    t_f = timeit(lambda: ops.mhsa_block_bwd(ds, qkv, lens, imgb, H, 12345, keep))
    t_a = timeit(lambda: ops.attn_core_bwd(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], lens, lens, None, ds, ref[..., :d], ref[..., d:2 * d], ref[..., 2 * d:], H, 12345, keep))
    t_g = timeit(lambda: ops.linear_backward_input(ref.view(B * T, 3 * d), wt, resid=ds.view(B * T, d)))
    print("B %d T %d keep %.1f: fused backward %.1f us; dmt_attn_bwd %.1f us + dx GEMM %.1f us = %.1f us" % (B, T, keep, t_f, t_a, t_g, t_a + t_g))

"""Time the short-sequence attention backward (dmt_attn_bwd: attn_bwd_co_kernel) at bench size with the attention dropout on and off,
against the forward.   python scripts/attn_bwd_time.py [T] [B]"""
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
qkv = torch.randn(B, T, 3 * d, device=dev).to(torch.bfloat16)
dout = torch.randn(B, T, d, device=dev).to(torch.bfloat16)
dqkv = torch.empty_like(qkv)
lens = torch.full((B,), T, dtype=torch.int32, device=dev)
q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
dq, dk, dv = dqkv[..., :d], dqkv[..., d:2 * d], dqkv[..., 2 * d:]
for keep in (1.0, 0.9):
    for _ in range(3):
        ops.attn_core_bwd(q, k, v, lens, lens, None, dout, dq, dk, dv, H, 12345, keep)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.attn_core_bwd(q, k, v, lens, lens, None, dout, dq, dk, dv, H, 12345, keep)
    e1.record()
    torch.cuda.synchronize()
    print("attention backward, B %d T %d, dropout keep %.1f: %.1f us per launch" % (B, T, keep, e0.elapsed_time(e1) * 50.0))

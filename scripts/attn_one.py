"""Self-attention forward + backward at bench size (B=4096, T=50, d=320, 4 heads), a few launches (for rocprofv3 --pmc)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cikm2020_dmt_amd import ops
dev = torch.device('cuda')
B, T, d, H = 4096, 50, 320, 4
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
qkv = torch.randn(B, T, 3 * d, device=dev).to(torch.bfloat16).requires_grad_(True)
x = torch.randn(B, T, d, device=dev).to(torch.bfloat16).requires_grad_(True)
lens = torch.full((B,), T, dtype=torch.int32, device=dev)
for _ in range(reps):
    out = ops.AttnFn.apply(qkv, None, x, lens, lens, H, d, True, 12345, 0.9)
    out.backward(torch.ones_like(out))
torch.cuda.synchronize()
print("done")

"""Times dmt_wgrad320 alone on the step's three shapes: python scripts/wgrad_micro.py [M]
   dW1 = x^T dh  (A [M,320], B [M,1280], bias of B)   dW2 = h^T ds (transposed form: A = ds [M,320], B = h [M,1280], bias of A)   dWqkv = x^T dqkv (N = 960)"""
import sys, torch
sys.path.insert(0, ".")
from cikm2020_dmt_amd import ops
M = int(sys.argv[1]) if len(sys.argv) > 1 else 204800
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
x = torch.randn(M, 320, generator=g).to(torch.bfloat16).to(dev)
for name, N, transposed, bias_of in (("dW1 (N=1280)", 1280, False, 1), ("dW2 (N=1280, transposed)", 1280, True, 2), ("dWqkv (N=960)", 960, False, 1)):
    b = torch.randn(M, N, generator=g).to(torch.bfloat16).to(dev)
    C = torch.zeros((N, 320) if transposed else (320, N), dtype=torch.float32, device=dev)
    bias = torch.zeros(320 if bias_of == 2 else N, dtype=torch.float32, device=dev)
    fn = lambda: ops.wgrad320(x, b, C, transposed, bias, bias_of)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print("%s M=%d: %.1f us  %.0f TF/s  %.2f TB/s of operands" % (name, M, us, 2.0 * M * 320 * N / us / 1e6, (M * 320 + M * N) * 2 / us / 1e6))

"""One GEMM form, a few launches (for rocprofv3 --pmc passes).  usage: gemm_one.py kk|mn [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cikm2020_dmt_amd import ops
dev = torch.device('cuda')
form = sys.argv[1]; reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
if form == 'kk':
    M, N, K = 204800, 1280, 320
    A = torch.randn(M, K, device=dev).to(torch.bfloat16); B = torch.randn(N, K, device=dev).to(torch.bfloat16)
    C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    f = lambda: ops.gemm(A, K, 1, B, 1, K, M, N, K, C, N)
else:
    M, N, K = 321, 1280, 204800
    x = torch.randn(K, 320, device=dev).to(torch.bfloat16); dy = torch.randn(K, N, device=dev).to(torch.bfloat16)
    f = lambda: ops.linear_backward_weight(x, dy, want_bias=True)
for _ in range(reps): f()
torch.cuda.synchronize()
print("done", form)

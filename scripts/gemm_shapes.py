"""Forward-form GEMM timing over N (K fixed): looks for shape-specific slow-downs in gemm_glds_kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cikm2020_dmt_amd import ops
dev = torch.device('cuda')
M = 204800
for (N, K, bias) in [(1280, 320, 0), (1280, 320, 1), (1024, 320, 0), (960, 320, 0), (960, 320, 1), (896, 320, 0), (640, 320, 0), (320, 1280, 0), (320, 960, 0), (384, 960, 0), (320, 320, 0)]:
    A = torch.randn(M, K, device=dev).to(torch.bfloat16); B = torch.randn(N, K, device=dev).to(torch.bfloat16)
    C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    bv = torch.randn(N, device=dev) if bias else None
    f = lambda: ops.gemm(A, K, 1, B, 1, K, M, N, K, C, N, bias=bv)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    tiles = (M // 128) * ((N + 127) // 128)
    print("N=%5d K=%5d bias=%d  %7.1f us  %6.1f TF/s  %5.2f ns/tile" % (N, K, bias, ms * 1e3, 2.0 * M * N * K / ms / 1e9, ms * 1e6 / tiles))

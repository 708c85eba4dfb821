"""What the vendor GEMM library (hipBLASLt / rocBLAS through torch.matmul) does on this step's plain-GEMM shapes, for comparison with the
hand-written kernels (bench.py lists their times): python scripts/blas_shapes.py"""
import torch
dev = torch.device("cuda:0")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
M = 204800
shapes = [("QKV projection        x[M,320] W[320,960]", (M, 320), (320, 960), False),
          ("QKV input gradient    dqkv[M,960] W^T[960,320]", (M, 960), (960, 320), False),
          ("FFN GEMM 1            x[M,320] W1[320,1280]", (M, 320), (320, 1280), False),
          ("FFN GEMM 2            h[M,1280] W2[1280,320]", (M, 1280), (1280, 320), False),
          ("FFN weight gradient   x^T[320,M] dh[M,1280]", (M, 320), (M, 1280), True),
          ("QKV weight gradient   x^T[320,M] dqkv[M,960]", (M, 320), (M, 960), True),
          ("MMoE layer 0          z[4096,1224] W[1224,2056]", (4096, 1224), (1224, 2056), False),
          ("decoder B-row GEMM    y[4096,320] W[320,320]", (4096, 320), (320, 320), False)]
for name, sa, sb, ta in shapes:
    a = torch.randn(sa, device=dev).to(torch.bfloat16)
    b = torch.randn(sb, device=dev).to(torch.bfloat16)
    if ta:
        fn = lambda: torch.matmul(a.t(), b)
        flops = 2.0 * sa[0] * sa[1] * sb[1]
    else:
        fn = lambda: torch.matmul(a, b)
        flops = 2.0 * sa[0] * sa[1] * sb[1]
    us = t(fn)
    print("%-52s %8.1f us  %7.0f TF/s" % (name, us, flops / us / 1e6))

"""GEMM microbenchmark: the dominant train-step shapes, timed with HIP events (and profiled with rocprofv3 --pmc)."""
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cikm2020_dmt_amd import ops
dev = torch.device('cuda')
shapes = [("ffn1_fwd", 204800, 1280, 320, 'kk'), ("ffn2_fwd", 204800, 320, 1280, 'kk'), ("qkv_fwd", 204800, 960, 320, 'kk'),
          ("ffn1_dW", 321, 1280, 204800, 'mn'), ("ffn2_dW", 1281, 320, 204800, 'mn'),
          ("mmoe_dW", 3048, 2056, 4096, 'mn'), ("mmoe_dW/s1", 3048, 2056, 4096, 'mn1'), ("mmoe_dW/s4", 3048, 2056, 4096, 'mn4'),
          ("exp1_dW", 513, 256, 4096, 'mn'), ("exp1_dW/s32", 513, 256, 4096, 'mn32'),
          ("dec_ffn1", 4096, 1280, 320, 'kk')]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for name, M, N, K, form in shapes:
    if form == 'kk':
        A = torch.randn(M, K, device=dev).to(torch.bfloat16); B = torch.randn(N, K, device=dev).to(torch.bfloat16)
        C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        f = lambda: ops.gemm(A, K, 1, B, 1, K, M, N, K, C, N)
    else:
        if len(form) > 2:
            ops._pick_split = (lambda sp: (lambda tiles, red: sp))(int(form[2:]))
        x = torch.randn(K, (M - 1 + 7) // 8 * 8, device=dev).to(torch.bfloat16)[:, :M - 1]; dy = torch.randn(K, N, device=dev).to(torch.bfloat16)
        f = lambda: ops.linear_backward_weight(x, dy, want_bias=True)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("%-10s M=%7d N=%5d K=%7d  %8.3f ms  %7.1f TF/s" % (name, M, N, K, ms, 2.0 * M * N * K / ms / 1e9))

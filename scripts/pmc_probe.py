import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
which = sys.argv[1]
dev = torch.device('cuda')
if which == 'torch':
    a = torch.randn(4096, 4096, device=dev); b = a @ a; torch.cuda.synchronize()
elif which == 'ln':
    from cikm2020_dmt_amd import ops
    x = torch.randn(4096, 320, device=dev).to(torch.bfloat16); g = torch.ones(320, device=dev); bb = torch.zeros(320, device=dev)
    y = ops.layer_norm(x, g, bb); torch.cuda.synchronize()
elif which == 'smallgemm':
    from cikm2020_dmt_amd import ops
    A = torch.randn(256, 128, device=dev).to(torch.bfloat16); B = torch.randn(128, 128, device=dev).to(torch.bfloat16)
    C = torch.empty(256, 128, dtype=torch.bfloat16, device=dev)
    ops.gemm(A, 128, 1, B, 1, 128, 256, 128, 128, C, 128); torch.cuda.synchronize()
print("done", which)

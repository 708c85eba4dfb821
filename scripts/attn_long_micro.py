"""Micro-benchmark of the attention core at long sequences: fused flash-style kernels (dmt_attn_long.hip) vs the unfused batched-GEMM form."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cikm2020_dmt_amd import ops

def run(B, Tq, Tk, H, dh, fused, drop, iters=10, fp8=False):
    ko = ops.KernelOptions(attn_long_fused=fused, attn_mma_fp8=fp8)
    d = H * dh
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(1)
    if Tq == Tk:
        qkv = (torch.randn((B, Tq, 3 * d), generator=g) * 0.7).to(torch.bfloat16).to(dev).requires_grad_(True)
        kv = None
    else:
        qkv = (torch.randn((B, Tq, d), generator=g) * 0.7).to(torch.bfloat16).to(dev).requires_grad_(True)
        kv = (torch.randn((B, Tk, 2 * d), generator=g) * 0.7).to(torch.bfloat16).to(dev).requires_grad_(True)
    x = torch.randn((B, Tq, d), generator=g).to(torch.bfloat16).to(dev)
    ql = torch.full((B,), Tq, dtype=torch.int32, device=dev)
    kl = torch.full((B,), Tk, dtype=torch.int32, device=dev)
    w = torch.randn((B, Tq, d), generator=g).to(torch.bfloat16).to(dev)
    def fwd():
        return ops.AttnFn.apply(qkv, kv, x, ql, kl, H, d, kv is None, 1234 if drop else 0, 0.9 if drop else 1.0, ko)
    for _ in range(2):
        out = fwd(); out.backward(w)
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for _ in range(iters):
        e[0].record(); out = fwd(); e[1].record(); out.backward(w); e[2].record()
        torch.cuda.synchronize()
        tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
    fl = 4.0 * B * H * Tq * Tk * dh
    print("B=%d Tq=%d Tk=%d H=%d dh=%d fused=%d drop=%d fp8=%d  fwd %.3f ms (%.1f TF/s)  bwd %.3f ms (%.1f TF/s)" %
          (B, Tq, Tk, H, dh, fused, drop, int(fp8), tf / iters, fl / (tf / iters) / 1e9, tb / iters, 2.5 * fl / (tb / iters) / 1e9))

if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    for fused in (1, 0):
        for drop in (0, 1):
            run(B, 200, 200, 4, 80, fused, drop)
    run(B, 1, 200, 4, 80, 1, 1)
    run(B, 1, 200, 4, 80, 0, 1)
    run(B, 128, 128, 4, 80, 1, 1)

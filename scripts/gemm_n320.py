"""dmt_gemm direct-to-LDS route on the QKV input-gradient shape: how much of N = 320 is the half-empty third column tile.
    python scripts/gemm_n320.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cikm2020_dmt_amd import ops          # noqa: E402


def main():
    dev = torch.device("cuda:0")
    M, K = 204800, 960
    x = (torch.randn((M, K), device=dev) * 0.1).to(torch.bfloat16)
    for N in (256, 320, 384, 64, 128):
        w = (torch.randn((N, K), device=dev) * 0.1).to(torch.bfloat16)
        r = (torch.randn((M, N), device=dev) * 0.1).to(torch.bfloat16)
        o = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        for resid in (None, r):
            kw = dict(resid=resid, ldr=N) if resid is not None else {}
            for _ in range(3):
                ops.gemm(x, K, 1, w, 1, K, M, N, K, o, N, **kw)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.gemm(x, K, 1, w, 1, K, M, N, K, o, N, **kw)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 100
            print("M=%d K=%d N=%3d resid=%d: %.1f us, %.0f TF/s" % (M, K, N, resid is not None, us, 2.0 * M * K * N / us / 1e6))


if __name__ == "__main__":
    main()

"""Times dmt_mhsa_block_fwd alone: python scripts/mhsa_micro.py [B] [T]"""
import sys, torch
sys.path.insert(0, ".")
from cikm2020_dmt_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
T = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
x = torch.randn(B, T, 320, generator=g).to(torch.bfloat16).to(dev)
w = (torch.randn(320, 960, generator=g) * 0.05).to(dev); b = torch.zeros(960, device=dev)
gamma = torch.ones(320, device=dev); beta = torch.zeros(320, device=dev)
lens = torch.full((B,), T, dtype=torch.int32, device=dev)
img = torch.empty(ops.mhsa_image_bytes(), dtype=torch.uint8, device=dev)
ops.mhsa_image_build(w, img)
for name, side, keep in (("train(dropout)", True, 0.9), ("train(no dropout)", True, 1.0), ("infer", False, 1.0)):
    fn = lambda: ops.mhsa_block_fwd(x, lens, img, b, gamma, beta, 1e-8, 4, 123, keep, want_side=side)
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    fl = 2.0 * B * T * 320 * 960 + 4.0 * B * T * T * 320
    print("%s B=%d T=%d: %.1f us  %.0f TF/s" % (name, B, T, us, fl / us / 1e6))

"""Times dmt_chain2 (fused ff + ln) alone: python scripts/chain_micro.py [M]   (DMT_CHAIN_DEBUG selects a timing variant)."""
import sys, torch
sys.path.insert(0, ".")
from cikm2020_dmt_amd import _lib as L, ops
M = int(sys.argv[1]) if len(sys.argv) > 1 else 204800
geo = (320, 1280, 320)
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
x = torch.randn(M, 320, generator=g).to(torch.bfloat16).to(dev)
w1 = (torch.randn(320, 1280, generator=g) * 0.05).to(dev); b1 = torch.zeros(1280, device=dev)
w2 = (torch.randn(1280, 320, generator=g) * 0.03).to(dev); b2 = torch.zeros(320, device=dev)
gamma = torch.ones(320, device=dev); beta = torch.zeros(320, device=dev)
n = ops.chain_image_bytes(*geo)
fwd = torch.empty(n, dtype=torch.uint8, device=dev); bwd = torch.empty(n, dtype=torch.uint8, device=dev)
ops.chain_image_build(geo, w1, 1, 1280, w2, 1, 320, b1, fwd)
ops.chain_image_build(geo, w2, 320, 1, w1, 1280, 1, None, bwd)
y = torch.empty_like(x); s = torch.empty_like(x); h = torch.empty(M, 1280, dtype=torch.bfloat16, device=dev)
st = torch.empty(M, 2, device=dev); mask = torch.zeros(4 * ((M + 127) // 128), 40, 64, dtype=torch.int16, device=dev)
dx = torch.empty_like(x); dh = torch.empty_like(h)
def f(): ops._chain_call(L.DMT_CHAIN_FFN_LN, geo, x, fwd, M, bias2=b2, gamma=gamma, beta=beta, eps=1e-8, s_out=s, y_out=y, stats=st, mid_out=h, mask=mask)
def fi(): ops._chain_call(L.DMT_CHAIN_FFN_LN, geo, x, fwd, M, bias2=b2, gamma=gamma, beta=beta, eps=1e-8, y_out=y)
def b(): ops._chain_call(L.DMT_CHAIN_FFN_BWD, geo, x, bwd, M, s_out=dx, mid_out=dh, mask=mask)
for name, fn in (("fwd(train)", f), ("fwd(infer)", fi), ("bwd", b)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print("%s M=%d: %.1f us  %.0f TF/s" % (name, M, us, 2.0 * M * 1280 * 640 / us / 1e6))

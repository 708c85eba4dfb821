"""Random-shape cross-check of the bf16 attention kernels (coalesced MFMA, single-query, legacy MFMA) against the scalar fp32
kernels of the same library: forward + all input gradients, self and cross attention, ragged lengths, dropout on/off."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cikm2020_dmt_amd import ops
dev = torch.device("cuda")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 150
worst = (0.0, None)
bad = []
for it in range(N):
    B = int(rng.integers(1, 10)); H = int(rng.choice([1, 2, 4])); dh = int(rng.choice([16, 20, 32, 64, 80]))
    self_attn = bool(rng.integers(0, 2))
    Tk = int(rng.integers(1, 65)) if rng.random() < 0.85 else int(rng.integers(65, 140))
    Tq = Tk if self_attn else int(rng.choice([1, int(rng.integers(1, 65))]))
    keep = float(rng.choice([1.0, 0.9, 0.5])); seed = int(rng.integers(1, 2 ** 31))
    d = H * dh
    g = torch.Generator().manual_seed(int(rng.integers(0, 2 ** 31)))
    ql = torch.tensor(rng.integers(1, Tq + 1, size=B), dtype=torch.int32, device=dev)
    kl = ql if self_attn else torch.tensor(rng.integers(1, Tk + 1, size=B), dtype=torch.int32, device=dev)
    valid = (torch.arange(Tq, device=dev)[None, :, None] < ql[:, None, None])
    w = torch.randn((B, Tq, d), generator=g).to(dev) * valid
    res = []
    for dt in (torch.float32, torch.bfloat16):
        if self_attn:
            a = (torch.randn((B, Tq, 3 * d), generator=torch.Generator().manual_seed(it)) * 0.6).to(torch.bfloat16).to(dt).to(dev).requires_grad_(True)
            b = None
        else:
            a = (torch.randn((B, Tq, d), generator=torch.Generator().manual_seed(it)) * 0.6).to(torch.bfloat16).to(dt).to(dev).requires_grad_(True)
            b = (torch.randn((B, Tk, 2 * d), generator=torch.Generator().manual_seed(it + 7)) * 0.6).to(torch.bfloat16).to(dt).to(dev).requires_grad_(True)
        x = torch.randn((B, Tq, d), generator=torch.Generator().manual_seed(it + 3)).to(torch.bfloat16).to(dt).to(dev)
        out = ops.AttnFn.apply(a, b, x, ql, kl, H, d, self_attn, seed, keep)
        (out.float() * w).sum().backward()
        res.append((out.detach().float(), a.grad.detach().float(), None if b is None else b.grad.detach().float()))
    eo = (((res[0][0] - res[1][0]).abs() * valid).max() / (res[0][0].abs() * valid).max().clamp_min(1e-6)).item()
    ega = ((res[0][1] - res[1][1]).abs().max() / res[0][1].abs().max().clamp_min(1e-6)).item()
    egb = 0.0 if res[0][2] is None else ((res[0][2] - res[1][2]).abs().max() / res[0][2].abs().max().clamp_min(1e-6)).item()
    e = max(eo, ega, egb)
    cfg = dict(B=B, H=H, dh=dh, Tq=Tq, Tk=Tk, self_attn=self_attn, keep=keep)
    if not np.isfinite(e) or e > 6e-2:
        bad.append((e, cfg))
    if e > worst[0]:
        worst = (e, cfg)
print("configs", N, "worst", worst, "bad", len(bad))
for b_ in bad[:10]:
    print("BAD", b_)

"""Random-shape sweep of the long-sequence attention kernels (dmt_attn_long.hip) against the unfused form of the same library:
Tq, Tk in 1..256 (at least one > 64), every instantiated head dim, ragged lengths INCLUDING 0 and T, dropout on / off, self and cross
attention, fp8 forward.  Prints the worst relative errors; exits non-zero past the tolerances of tests/test_gpu_attn_long.py."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cikm2020_dmt_amd import ops

def run(pq, pkv, x, ql, kl, w, H, drop, fused, fp8=False):
    d = x.shape[2]
    a = pq.clone().requires_grad_(True)
    b = pkv.clone().requires_grad_(True) if pkv is not None else None
    xd = x.clone().requires_grad_(True)
    out = ops.AttnFn.apply(a, b, xd, ql, kl, H, d, pkv is None, 0x1234567 if drop else 0, 0.9 if drop else 1.0,
                           ops.KernelOptions(attn_mma_fp8=fp8, attn_long_fused=fused))
    (out.float() * w).sum().backward()
    return out.detach().float(), [a.grad.float()] + ([b.grad.float()] if b is not None else []) + [xd.grad.float()]

def main(n_cases=400, seed=0):
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda")
    worst = dict(out=0.0, grad=0.0, pad=0.0, f8=0.0)
    for it in range(n_cases):
        dh = int(rng.choice([16, 32, 64, 80])); H = int(rng.choice([1, 2, 4])); d = H * dh
        B = int(rng.integers(1, 6))
        self_attn = rng.random() < 0.5
        Tk = int(rng.integers(65, 257)) if rng.random() < 0.7 else int(rng.integers(1, 257))
        Tq = Tk if self_attn else int(rng.integers(1, 257))
        if max(Tq, Tk) <= 64:
            Tk = int(rng.integers(65, 257)); Tq = Tk if self_attn else Tq
        drop = bool(rng.random() < 0.5)
        g = torch.Generator(device="cpu").manual_seed(int(rng.integers(1 << 30)))
        if self_attn:
            pq = (torch.randn((B, Tq, 3 * d), generator=g) * 0.7).to(torch.bfloat16).to(dev); pkv = None
        else:
            pq = (torch.randn((B, Tq, d), generator=g) * 0.7).to(torch.bfloat16).to(dev)
            pkv = (torch.randn((B, Tk, 2 * d), generator=g) * 0.7).to(torch.bfloat16).to(dev)
        x = torch.randn((B, Tq, d), generator=g).to(torch.bfloat16).to(dev)
        def lens(T):
            l = rng.integers(0, T + 1, size=B)
            if rng.random() < 0.3: l[rng.integers(B)] = T
            if rng.random() < 0.15: l[rng.integers(B)] = 0
            return torch.tensor(l, dtype=torch.int32, device=dev)
        ql = lens(Tq); kl = ql if self_attn else lens(Tk)
        valid = (torch.arange(Tq, device=dev)[None, :, None] < ql[:, None, None])
        w = torch.randn((B, Tq, d), generator=g).to(dev) * valid
        o1, g1 = run(pq, pkv, x, ql, kl, w, H, drop, True)
        o0, g0 = run(pq, pkv, x, ql, kl, w, H, drop, False)
        sc = max((o0.abs() * valid).max().item(), 1e-6)
        eo = ((o1 - o0).abs() * valid).max().item() / sc
        pad = ~valid.expand_as(o0)
        ep = ((o1 - o0).abs()[pad] / (o0.abs()[pad] + 1.0)).max().item() if pad.any() else 0.0
        eg = max((a - b).abs().max().item() / (b.abs().max().item() + 1e-20) for a, b in zip(g1, g0))
        bad = (not torch.isfinite(o1).all()) or any(not torch.isfinite(t).all() for t in g1)
        e8 = 0.0
        if Tq > 1 and rng.random() < 0.3:
            x0 = torch.zeros_like(x)
            b0, _ = run(pq, pkv, x0, ql, kl, w, H, drop, True)
            b8, _ = run(pq, pkv, x0, ql, kl, w, H, drop, True, fp8=True)
            e8 = ((b8 - b0).abs() * valid).max().item() / max((b0.abs() * valid).max().item(), 1e-6)
            bad = bad or not torch.isfinite(b8 * valid).all()
        for k, v in (("out", eo), ("grad", eg), ("pad", ep), ("f8", e8)):
            worst[k] = max(worst[k], v)
        if bad or eo > 2e-2 or eg > 4e-2 or ep > 3e-2 or e8 > 0.12:
            print("FAIL case", it, dict(B=B, Tq=Tq, Tk=Tk, H=H, dh=dh, drop=drop, self_attn=self_attn, ql=ql.tolist(), kl=kl.tolist()), eo, eg, ep, e8, bad)
            sys.exit(1)
    print("cases", n_cases, "worst", worst)

if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 400, int(sys.argv[2]) if len(sys.argv) > 2 else 0)

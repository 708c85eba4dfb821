"""Random-shape cross-check of dmt_gemm's bf16 paths (direct-to-LDS forward, weight-gradient with both stage depths, generic)
against torch fp32 matmul on the same bf16 inputs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cikm2020_dmt_amd import ops
dev = torch.device("cuda")
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N_ = int(sys.argv[2]) if len(sys.argv) > 2 else 60
worst, bad = (0.0, None), []
for it in range(N_):
    form = rng.choice(["kk", "mn"])
    if form == "kk":
        M = int(rng.choice([1, 77, 128, 1000, 4096, 20000])); N = int(rng.choice([8, 64, 320, 960, 1280, 2056])); K = int(rng.choice([64, 320, 1280, 3047, 72, 960]))
        A = (torch.randn(M, K, device=dev) * 0.5).to(torch.bfloat16); Bm = (torch.randn(N, K, device=dev) * 0.5).to(torch.bfloat16)
        bias = torch.randn(N, device=dev) if rng.random() < 0.5 else None
        relu = bool(rng.integers(0, 2)) and bias is not None
        C = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        kw = {}
        gate = resid = None
        if not relu and rng.random() < 0.4:
            gate = torch.randn(M, N, device=dev).to(torch.bfloat16); kw.update(gate=gate, ldg=N)
        if rng.random() < 0.4:
            resid = torch.randn(M, N, device=dev).to(torch.bfloat16); kw.update(resid=resid, ldr=N)
        ops.gemm(A, K, 1, Bm, 1, K, M, N, K, C, N, bias=bias, act_ncols=N if relu else 0, **kw)
        ref = A.float() @ Bm.float().t() + (bias if bias is not None else 0)
        if relu:
            ref = torch.relu(ref)
        if gate is not None:
            ref = torch.where(gate.float() > 0, ref, torch.zeros_like(ref))
        if resid is not None:
            ref = ref + resid.float()
        got = C.float()
        tol = 2e-2
    else:
        Kin = int(rng.choice([16, 320, 1280, 513, 3047])); N = int(rng.choice([32, 320, 960, 1280])); R = int(rng.choice([64, 4096, 4160, 8192, 40960, 65600]))
        x = (torch.randn(R, (Kin + 7) // 8 * 8, device=dev) * 0.5).to(torch.bfloat16)[:, :Kin]; dy = (torch.randn(R, N, device=dev) * 0.5).to(torch.bfloat16)
        dW, db = ops.linear_backward_weight(x, dy, want_bias=True)
        ref = torch.cat([x.float().t() @ dy.float(), dy.float().sum(0, keepdim=True)], 0)
        got = torch.cat([dW, db[None, :]], 0)
        M, K = Kin + 1, R
        tol = 3e-3
    e = ((got - ref).abs().max() / ref.abs().max().clamp_min(1e-6)).item()
    cfg = (form, M, N, K)
    if not np.isfinite(e) or e > tol:
        bad.append((e, cfg))
    if e > worst[0]:
        worst = (e, cfg)
print("configs", N_, "worst", worst, "bad", bad[:8])

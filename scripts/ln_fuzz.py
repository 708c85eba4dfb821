import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from cikm2020_dmt_amd import ops
dev = torch.device("cuda")
rng = np.random.default_rng(0)
worst = (0, None)
for it in range(120):
    rows = int(rng.integers(1, 6000)); d = int(rng.choice([7, 8, 16, 20, 64, 80, 100, 320, 328, 512, 1024]))
    dt = torch.bfloat16 if rng.random() < 0.6 else torch.float32
    x = (torch.randn(rows, d, device=dev) * 2 + 0.3).to(dt).requires_grad_(True)
    g = (1 + 0.1 * torch.randn(d, device=dev)).requires_grad_(True); b = (0.1 * torch.randn(d, device=dev)).requires_grad_(True)
    y = ops.layer_norm(x, g, b)
    w = torch.randn(rows, d, device=dev)
    (y.float() * w).sum().backward()
    xr = x.detach().float().requires_grad_(True); gr = g.detach().clone().requires_grad_(True); br = b.detach().clone().requires_grad_(True)
    mu = xr.mean(-1, keepdim=True); var = ((xr - mu) ** 2).mean(-1, keepdim=True)
    yr = gr * (xr - mu) / torch.sqrt(var + 1e-8) + br        # TransformerModel_util.ln: epsilon inside the sqrt
    (yr * w).sum().backward()
    tol = 2e-2 if dt == torch.bfloat16 else 2e-4
    errs = [((y.float() - yr).abs().max() / yr.abs().max()).item(), ((x.grad.float() - xr.grad).abs().max() / xr.grad.abs().max()).item(),
            ((g.grad - gr.grad).abs().max() / gr.grad.abs().max().clamp_min(1e-6)).item(), ((b.grad - br.grad).abs().max() / br.grad.abs().max().clamp_min(1e-6)).item()]
    e = max(errs) / tol
    if e > worst[0]:
        worst = (e, (rows, d, str(dt), errs))
print("worst (in units of tolerance)", worst)

"""Fused expert-MLP + gate kernels (dmt_mmoe_experts_fwd/bwd) against the launch-per-layer form of the same engine
(batched dmt_gemm per expert layer + dmt_mmoe_mix_*: independent code) and against the fp64 oracle's expert_gate."""
import numpy as np
import pytest
import torch

from oracle import dmt_oracle as O
from cikm2020_dmt_amd import ops
from cikm2020_dmt_amd import spec as S
from cikm2020_dmt_amd.engine import DMTEngine
from cikm2020_dmt_amd.variables import VariableStore

pytestmark = pytest.mark.gpu


def _engine(cuda, seed=3):
    sp = S.scaled_spec(S.e64_spec(), {"Sku": 2000, "Brand": 300, "Shopid": 300, "Cid3": 120})
    st = VariableStore(sp, cuda, torch.bfloat16, seed=seed)
    return sp, st, DMTEngine(sp, st)


@pytest.mark.parametrize("B", [1, 63, 64, 200, 4096])
def test_fused_expert_kernels_match_the_per_layer_form(cuda, B):
    sp, st, eng = _engine(cuda)
    K = eng.plan.K
    g = torch.Generator(device="cpu").manual_seed(B)
    z0 = (torch.randn((B, K), generator=g) * 0.5).to(torch.bfloat16).to(cuda)
    w = [torch.randn((B, sp["hidden_units_bottom"][-1]), generator=g).to(cuda) for _ in range(sp["num_tasks"])]
    assert ops.mmoe_experts_supported(sp["hidden_units_bottom"], sp["num_experts"], sp["num_tasks"], torch.bfloat16)
    res = []
    for fused in (True, False):
        eng.use_mmoe_fused = fused
        st.zero_grad()
        z = z0.clone().requires_grad_(True)
        tasks = eng.expert_gate(z)
        gates = eng.intermediates["gates"].clone()
        loss = sum((t.float() * wi).sum() for t, wi in zip(tasks, w))
        loss.backward()
        res.append(([t.detach().float() for t in tasks], gates, z.grad.float().clone(), st.grads.clone()))
    (t1, g1, dz1, gr1), (t0, g0, dz0, gr0) = res
    assert torch.allclose(g1, g0, atol=1e-6)
    for a, b in zip(t1, t0):
        assert (a - b).abs().max().item() <= 2e-2 * b.abs().max().item() + 1e-3
    assert (dz1 - dz0).abs().max().item() <= 2e-2 * dz0.abs().max().item() + 1e-4
    # parameter gradients: every dense leaf the MMoE bottom touches (experts, gates, layer 0), one flat arena
    touched = gr0 != 0
    assert int(touched.sum()) > 1e6 and bool(((gr1 != 0) == touched).all())
    err = (gr1 - gr0).abs().max().item() / gr0.abs().max().item()
    assert err < 2e-2, err
    rel = ((gr1 - gr0).norm() / gr0.norm()).item()
    assert rel < 5e-3, rel


def test_fused_expert_kernels_match_the_oracle(cuda):
    sp, st, eng = _engine(cuda, seed=5)
    so = dict(sp)
    P = st.state_dict()
    B, K = 37, eng.plan.K
    rng = np.random.default_rng(2)
    z = (rng.standard_normal((B, K)) * 0.5).astype(np.float32)
    zd = torch.tensor(z).to(torch.bfloat16).to(cuda)
    eng.use_mmoe_fused = True
    with torch.no_grad():
        tasks = eng.expert_gate(zd)
    want, want_gates = O.expert_gate(zd.float().cpu().numpy().astype(np.float64), P, so)
    got_gates = eng.intermediates["gates"].cpu().numpy()
    for t in range(len(want_gates)):
        assert np.abs(got_gates[t] - want_gates[t]).max() < 2e-2         # (bf16 logits)
    for a, b in zip(tasks, want):
        e = np.abs(a.float().cpu().numpy() - b).max() / (np.abs(b).max() + 1e-9)
        assert e < 3e-2, e

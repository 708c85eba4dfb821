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


@pytest.mark.parametrize("B", [1, 31, 33, 257, 4096])
def test_split_expert_kernels_are_bit_identical_to_the_one_workgroup_form(cuda, B):
    """One workgroup per (row tile, expert) + a small second launch that sums the mixtures / finishes the gate-logit gradient in expert
    order (workspace form, round 4) against one workgroup per row tile walking its experts: same operations, same order -- every output
    bit for bit; gate_dx is exactly a multiplication of the expert-input columns by (g1 > 0).  Route asserted."""
    from cikm2020_dmt_amd import _lib as L
    sp, st, eng = _engine(cuda, seed=11)
    E, T, units = sp["num_experts"], sp["num_tasks"], sp["hidden_units_bottom"]
    g = torch.Generator(device="cpu").manual_seed(B + 5)
    ncol = E * units[0] + T * E
    g1v = (torch.randn((B, ncol), generator=g) * 0.7).to(torch.bfloat16).to(cuda)
    g1v[:, : E * units[0]].clamp_(min=0)              # (relu'd layer-0 outputs: about half of them zero)
    dmix = (torch.randn((T, B, units[-1]), generator=g) * 0.3).to(torch.bfloat16).to(cuda)
    names = [["mmoe_layers/expert-%d/expert-layer-%d/" % (e, li) for e in range(E)] for li in (1, 2)]
    args = ([eng._w(n + "weights") for n in names[0]], [eng._w(n + "weights") for n in names[1]],
            [eng._lf(n + "weights") for n in names[0]], [eng._lf(n + "biases") for n in names[0]],
            [eng._lf(n + "weights") for n in names[1]], [eng._lf(n + "biases") for n in names[1]], E, T)
    out = {}
    for key, gate_dx, split in (("one", False, False), ("split", False, True), ("split_gated", True, True)):
        st.zero_grad()
        x = g1v.clone().requires_grad_(True)
        with L.route_trace() as rt:
            mix, gates = ops.MmoeExpertsFn.apply(x, *args, gate_dx, split)
            mix.backward(dmix)
            torch.cuda.synchronize()
        want = "(split)" if split else ""
        assert rt.counts.get("dmt_mmoe_experts_fwd" + want, 0) == 1 and rt.counts.get("dmt_mmoe_experts_bwd" + want, 0) == 1, rt.counts
        out[key] = (mix.detach().clone(), gates.clone(), x.grad.clone(), st.grads.clone())
    for i, name in enumerate(("mix", "gates", "d g1")):
        assert torch.equal(out["one"][i], out["split"][i]), name
    # (the weight gradients are batched GEMMs over the saved d h1 / d h2 -- bit-identical operands; at large B they sum split-K pieces
    #  with fp32 atomics, whose order is not reproducible from run to run)
    ga, gb = out["one"][3], out["split"][3]
    assert ((ga - gb).norm() / ga.norm()).item() < 1e-6
    want = out["split"][2].clone()
    want[:, : E * units[0]] *= (g1v[:, : E * units[0]] > 0).to(want.dtype)
    assert torch.equal(out["split_gated"][2], want)
    assert torch.equal(out["split_gated"][0], out["split"][0])

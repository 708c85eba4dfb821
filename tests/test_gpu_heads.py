"""Fused heads kernels (dmt_heads_fwd/bwd: task towers + position-bias tower) against the layer-per-launch form of the same engine
(dmt_gemm + relu / dropout kernels: independent code), forward and every gradient, dropout on and off."""
import numpy as np
import pytest
import torch

from cikm2020_dmt_amd import ops
from cikm2020_dmt_amd import spec as S
from cikm2020_dmt_amd.engine import DMTEngine
from cikm2020_dmt_amd.variables import VariableStore

pytestmark = pytest.mark.gpu


def _engine(cuda, seed=3):
    sp = S.scaled_spec(S.e64_spec(), {"Sku": 2000, "Brand": 300, "Shopid": 300, "Cid3": 120})
    st = VariableStore(sp, cuda, torch.bfloat16, seed=seed)
    # non-zero biases: the initialiser leaves them at 0, which would hide a wrong bias index
    with torch.no_grad():
        g = torch.Generator(device="cpu").manual_seed(seed)
        for name, leaf in st.leaf.items():
            if leaf.dim() == 1 and ("bias" in name):
                leaf.copy_((torch.randn(leaf.shape, generator=g) * 0.1).to(leaf.device))
    st.refresh_shadows()
    return sp, st, DMTEngine(sp, st)


@pytest.mark.parametrize("drop", [False, True])
@pytest.mark.parametrize("B", [1, 63, 64, 300, 4096])
def test_fused_heads_match_the_layer_path(cuda, B, drop):
    sp, st, eng = _engine(cuda)
    T, U = sp["num_tasks"], sp["hidden_units_bottom"][-1]
    g = torch.Generator(device="cpu").manual_seed(B)
    mix0 = (torch.randn((T, B, U), generator=g) * 0.5).to(torch.bfloat16).to(cuda)
    zfull = (torch.randn((B, eng.plan.ldz), generator=g) * 0.5).to(torch.bfloat16).to(cuda)
    wl = [torch.randn((B, 1), generator=g).to(cuda) for _ in range(T + 1)]
    assert ops.heads_supported(U, sp["hidden_units_task"], eng.plan.bias_width, sp["hidden_units_bias"], T, torch.bfloat16)
    eng.dropout_step_seed = 77 if drop else None
    res = []
    for fused in (True, False):
        st.zero_grad()
        mix = mix0.clone().requires_grad_(True)
        z = zfull.clone().requires_grad_(True)
        zb = z[:, eng.plan.bias_off: eng.plan.bias_off + eng.plan.bias_width]
        if fused:
            (c, o), yb = eng.heads(mix, zb)
        else:
            tasks = list(ops.Unbind0Fn.apply(mix))
            c, o = (eng.build_tower(m, nm) for m, nm in zip(tasks, ("click", "order")))
            yb = eng.embedding_mlp_bias(zb)
        outs = (c, o, yb)
        sum((x.float() * w).sum() for x, w in zip(outs, wl)).backward()
        res.append(([x.detach().float().clone() for x in outs], mix.grad.float().clone(), z.grad.float().clone(), st.grads.clone()))
    (o1, dm1, dz1, gr1), (o0, dm0, dz0, gr0) = res
    for a, b in zip(o1, o0):
        assert a.shape == b.shape == (B, 1)
        assert (a - b).abs().max().item() <= 2e-2 * b.abs().max().item() + 2e-3, (a - b).abs().max().item()
    assert (dm1 - dm0).abs().max().item() <= 3e-2 * dm0.abs().max().item() + 1e-4
    assert (dz1 - dz0).abs().max().item() <= 3e-2 * dz0.abs().max().item() + 1e-4
    assert float(dz1[:, : eng.plan.bias_off].abs().max()) == 0.0                    # only the bias slice of z gets a gradient here
    touched = gr0 != 0
    assert int(touched.sum()) > 1000
    # (an exact zero can differ where a relu / dropout gate sits on a rounding boundary: compare values, not the zero pattern)
    err = (gr1 - gr0).abs().max().item() / gr0.abs().max().item()
    assert err < 3e-2, err
    rel = ((gr1 - gr0).norm() / gr0.norm()).item()
    assert rel < 1e-2, rel


def test_fused_heads_in_the_model_step(cuda):
    """The whole model with and without the fused heads: same loss and gradients to bf16 rounding (dropout on: same counters)."""
    from cikm2020_dmt_amd.data_feed.synthetic import make_batch
    from cikm2020_dmt_amd.train import Trainer
    sp = S.scaled_spec(S.e64_spec(), {"Sku": 2000, "Brand": 300, "Shopid": 300, "Cid3": 120})
    inputs, mask, _ = make_batch(sp, 48, seed=4, lengths="ragged", weights="random")
    out = []
    for fused in (True, False):
        tr = Trainer(sp, device=cuda, compute_dtype=torch.bfloat16, seed=9, dropout=True)
        tr.engine.use_heads_fused = fused
        loss = tr.forward_backward(tr.make_batch(inputs, mask))
        out.append((float(loss), tr.store.grads.clone()))
    assert abs(out[0][0] - out[1][0]) < 2e-3 * abs(out[1][0]) + 1e-3
    rel = ((out[0][1] - out[1][1]).norm() / out[1][1].norm()).item()
    assert rel < 2e-2, rel

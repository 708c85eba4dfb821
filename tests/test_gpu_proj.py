"""dmt_proj: the QKV projection with the weights streamed as an LDS image (dmt_chain.hip proj_kernel) against fp64 math on the bf16
operands, ragged row counts included, and end to end through SelfAttnBlockFn against the tiled-GEMM path."""
import numpy as np
import pytest
import torch

from cikm2020_dmt_amd import ops

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M", [1, 31, 128, 129, 4096 + 77, 128 * 513 + 5])
def test_proj_matches_fp64(cuda, M):
    g = torch.Generator(device="cpu").manual_seed(M)
    kin, n = 320, 960
    w = (torch.randn(kin, n, generator=g) * 0.05).to(cuda)
    b = (torch.randn(n, generator=g) * 0.1).to(cuda)
    x = (torch.randn(M + 3, kin + 8, generator=g)).to(cuda).to(torch.bfloat16)[1: M + 1, :kin]        # strided, offset rows
    assert x.stride(0) == kin + 8
    img = torch.empty(ops.proj_image_bytes(kin, n), dtype=torch.uint8, device=cuda)
    ops.proj_image_build(w, b, img)
    W = ops.Weight(w)
    W.proj = img
    assert ops.proj_ok(x, W)
    out = ops.proj_forward(x, W, n)
    torch.cuda.synchronize()
    ref = x.double() @ w.to(torch.bfloat16).double() + b.double()
    err = (out.double() - ref).abs()
    tol = 2 ** -8 * ref.abs() + 1e-3            # one bf16 rounding of the result + fp32 accumulation over 320 terms
    assert bool((err <= tol).all()), float((err - tol).max())


def test_proj_image_follows_transposed_views(cuda):
    """The image builder takes any strides: a transposed weight view gives the same image as its contiguous copy."""
    g = torch.Generator(device="cpu").manual_seed(3)
    wt = (torch.randn(960, 320, generator=g) * 0.05).to(cuda)
    nb = ops.proj_image_bytes(320, 960)
    a, b = torch.zeros(nb, dtype=torch.uint8, device=cuda), torch.zeros(nb, dtype=torch.uint8, device=cuda)
    ops.proj_image_build(wt.t(), None, a)
    ops.proj_image_build(wt.t().contiguous(), None, b)
    torch.cuda.synchronize()
    assert torch.equal(a, b)


def test_self_attention_block_same_with_and_without_proj(cuda):
    """Encoder self-attention at E64 width: forward output and all gradients with the streamed projection against the tiled GEMM."""
    from cikm2020_dmt_amd import spec as S
    from cikm2020_dmt_amd.data_feed.synthetic import make_batch
    from cikm2020_dmt_amd.train import Trainer
    sp = S.scaled_spec(S.e64_spec(), {"Sku": 20000, "Brand": 3000, "Shopid": 3000, "Cid3": 1200})
    inputs, mask, label = make_batch(sp, 96, seed=11, lengths="ragged", weights="random")
    res = []
    ops.set_deterministic(True)
    try:
        for use in (False, True):
            tr = Trainer(sp, device=cuda, compute_dtype=torch.bfloat16, seed=4, dropout=True)
            tr.engine.kopts = tr.engine.kopts.replace(use_proj=use)
            assert len(tr.store.proj) == 3
            loss = tr.forward_backward(tr.make_batch(inputs, mask, label))
            torch.cuda.synchronize()
            res.append((float(loss), tr.store.grads.clone()))
    finally:
        ops.set_deterministic(False)
    (l0, g0), (l1, g1) = res
    # same bf16 operands, same fp32 accumulation up to its order: the q | k | v values differ by rounding-level amounts at most
    assert abs(l0 - l1) < 2e-3 * abs(l0)
    assert ((g0 - g1).norm() / g0.norm()).item() < 2e-2

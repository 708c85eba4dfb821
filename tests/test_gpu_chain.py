"""dmt_chain2 (fused ff + ln, and its input gradient) against a plain PyTorch fp32 statement of the same op.

Reference: ff(inputs, [d_ff, d_model]) + ln() of /root/reference/DMT_code/model/net/TransformerModel_util.py:212-235, 58-78.
Inputs and weights are bf16 values (what the kernel reads); the reference accumulates in fp32 and rounds the d_ff-wide
activation to bf16 where the kernel does (it is an MFMA operand of the second GEMM).  Tolerances: outputs are bf16, so
|d| <= 2^-7 relative to the row scale (one bf16 ulp of the largest element plus accumulation-order noise).
"""
import numpy as np
import pytest
import torch

from cikm2020_dmt_amd import _lib as L
from cikm2020_dmt_amd import ops

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _mk(cuda, geo, M, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    kin, nmid, nout = geo
    x = (torch.randn(M, kin, generator=g) * 1.5).to(BF)
    w1 = (torch.randn(kin, nmid, generator=g) * (1.0 / kin) ** 0.5)
    b1 = torch.randn(nmid, generator=g) * 0.2
    w2 = (torch.randn(nmid, nout, generator=g) * (1.0 / nmid) ** 0.5)
    b2 = torch.randn(nout, generator=g) * 0.2
    gamma = 1.0 + 0.2 * torch.randn(nout, generator=g)
    beta = 0.1 * torch.randn(nout, generator=g)
    return [t.to(cuda) for t in (x, w1, b1, w2, b2, gamma, beta)]


def _images(geo, w1, b1, w2):
    n = ops.chain_image_bytes(*geo)
    assert n is not None
    fwd = torch.empty(n, dtype=torch.uint8, device=w1.device)
    bwd = torch.empty(n, dtype=torch.uint8, device=w1.device)
    ops.chain_image_build(geo, w1, 1, w1.stride(0), w2, 1, w2.stride(0), b1, fwd)
    ops.chain_image_build(geo, w2, w2.stride(0), 1, w1, w1.stride(0), 1, None, bwd)
    return fwd, bwd


def _ref_fwd(x, w1, b1, w2, b2, gamma, beta, eps=1e-8):
    xf, w1b, w2b = x.float(), w1.to(BF).float(), w2.to(BF).float()
    h = torch.relu(xf @ w1b + b1)
    hb = h.to(BF).float()
    s = hb @ w2b + b2 + xf
    mean = s.mean(-1, keepdim=True)
    var = ((s - mean) ** 2).mean(-1, keepdim=True)
    y = gamma * (s - mean) / torch.sqrt(var + eps) + beta
    return h, s, y, mean[:, 0], 1.0 / torch.sqrt(var + eps)[:, 0]


@pytest.mark.parametrize("geo", [(80, 320, 80), (320, 1280, 320)])
@pytest.mark.parametrize("M", [1, 31, 128, 129, 1000, 4099])
def test_ffn_ln_forward_and_input_gradient(cuda, geo, M):
    x, w1, b1, w2, b2, gamma, beta = _mk(cuda, geo, M, seed=100 + M)
    fwd, bwd = _images(geo, w1, b1, w2)
    kin, nmid, nout = geo
    y = torch.empty((M, nout), dtype=BF, device=cuda)
    s = torch.empty((M, nout), dtype=BF, device=cuda)
    h = torch.empty((M, nmid), dtype=BF, device=cuda)
    stats = torch.empty((M, 2), dtype=torch.float32, device=cuda)
    mask = torch.zeros((4 * ((M + 127) // 128), nmid // 32, 64), dtype=torch.int16, device=cuda)
    ops._chain_call(L.DMT_CHAIN_FFN_LN, geo, x, fwd, M, bias2=b2, gamma=gamma, beta=beta, eps=1e-8, s_out=s, y_out=y, stats=stats, mid_out=h,
                    mask=mask)
    torch.cuda.synchronize()
    h_ref, s_ref, y_ref, mean_ref, rstd_ref = _ref_fwd(x, w1, b1, w2, b2, gamma, beta)
    tol = 2.0 ** -7

    def rowscale(t):
        return t.abs().amax(dim=-1, keepdim=True).clamp_min(1e-3)

    assert ((h.float() - h_ref).abs() / rowscale(h_ref)).max() < tol
    assert ((s.float() - s_ref).abs() / rowscale(s_ref)).max() < tol
    assert ((y.float() - y_ref).abs() / rowscale(y_ref)).max() < 2 * tol
    assert (stats[:, 0] - mean_ref).abs().max() < 2e-2 * s_ref.abs().max()
    assert ((stats[:, 1] - rstd_ref).abs() / rstd_ref).max() < 2e-2
    # inference form (no side outputs) gives the same y
    y2 = torch.empty_like(y)
    ops._chain_call(L.DMT_CHAIN_FFN_LN, geo, x, fwd, M, bias2=b2, gamma=gamma, beta=beta, eps=1e-8, y_out=y2)
    assert torch.equal(y, y2)

    # ---- input gradient with the forward's gate bits: dh = (ds W2^T) * [h > 0], dx = dh W1^T + ds
    g = torch.Generator(device="cpu").manual_seed(7 + M)
    ds = torch.randn(M, nout, generator=g).to(BF).to(cuda)
    dx = torch.empty((M, kin), dtype=BF, device=cuda)
    dh = torch.empty((M, nmid), dtype=BF, device=cuda)
    ops._chain_call(L.DMT_CHAIN_FFN_BWD, geo, ds, bwd, M, s_out=dx, mid_out=dh, mask=mask)
    torch.cuda.synchronize()
    gate = (h.float() > 0).float()           # the kernel's own gate (h == 0 exactly where relu cut)
    dh_ref = (ds.float() @ w2.to(BF).float().t()) * gate
    dx_ref = dh_ref.to(BF).float() @ w1.to(BF).float().t() + ds.float()
    assert ((dh.float() - dh_ref).abs() / rowscale(dh_ref)).max() < tol
    assert ((dx.float() - dx_ref).abs() / rowscale(dx_ref)).max() < tol


def test_chain_autograd_function_matches_unfused_path(cuda):
    """FFNLNChainFn (one launch + LN gradient + chain backward) against FFNFn + LNFn (GEMM launches) on the same leaves."""
    geo = (320, 1280, 320)
    M = 777
    x, w1, b1, w2, b2, gamma, beta = _mk(cuda, geo, M, seed=3)
    fwd, bwd = _images(geo, w1, b1, w2)
    chain = dict(geo=geo, fwd=fwd, bwd=bwd)
    leaves = [t.clone().requires_grad_(True) for t in (w1, b1, w2, b2, gamma, beta)]
    xa = x.clone().requires_grad_(True)
    ya = ops.FFNLNChainFn.apply(xa, leaves[0], leaves[1], leaves[2], leaves[3], leaves[4], leaves[5], chain, 1e-8)
    g = torch.Generator(device="cpu").manual_seed(11)
    dy = torch.randn(M, 320, generator=g).to(BF).to(cuda)
    ya.backward(dy)
    leaves_b = [t.clone().requires_grad_(True) for t in (w1, b1, w2, b2, gamma, beta)]
    xb = x.clone().requires_grad_(True)
    wt1 = ops.Weight(w1, w1.to(BF), w1.to(BF).t().contiguous())
    wt2 = ops.Weight(w2, w2.to(BF), w2.to(BF).t().contiguous())
    sb = ops.FFNFn.apply(xb, leaves_b[0], leaves_b[1], leaves_b[2], leaves_b[3], wt1, wt2)
    yb = ops.layer_norm(sb, leaves_b[4], leaves_b[5])
    yb.backward(dy)
    assert (ya.float() - yb.float()).abs().max() < 0.05
    assert (xa.grad.float() - xb.grad.float()).abs().max() / xb.grad.float().abs().max() < 2e-2
    for a, b in zip(leaves, leaves_b):
        assert (a.grad - b.grad).abs().max() / b.grad.abs().max() < 2e-2

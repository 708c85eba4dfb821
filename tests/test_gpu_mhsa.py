"""dmt_mhsa_block_fwd (QKV projection + masked softmax attention + residual + LayerNorm in one launch) against
 (a) a plain PyTorch fp32 statement of multihead_attention + ln (TransformerModel_util.py:160-209, 11-56, 80-108, 58-78), incl. the
     key mask before / query mask after the softmax (-2^32 + 1) and the counter dropout of oracle/dmt_oracle.py, and
 (b) the unfused HIP path (GEMM + attention + LN kernels) through autograd, forward and every gradient.
Tolerance: bf16 outputs, |d| <= 2^-6 of the row scale for live rows (the scores pass through bf16 Q, K, P).
"""
import numpy as np
import pytest
import torch

from cikm2020_dmt_amd import ops
from cikm2020_dmt_amd._lib import route_trace as L_route
from oracle import dmt_oracle as O

pytestmark = pytest.mark.gpu
BF = torch.bfloat16
PAD = -4294967295.0


def _mk(cuda, B, T, seed):
    g = torch.Generator().manual_seed(seed)
    d = 320
    x = (torch.randn(B, T, d, generator=g)).to(BF)
    w = torch.randn(d, 3 * d, generator=g) * (1.0 / d) ** 0.5
    b = torch.randn(3 * d, generator=g) * 0.1
    gamma = 1.0 + 0.2 * torch.randn(d, generator=g)
    beta = 0.1 * torch.randn(d, generator=g)
    lens = torch.randint(1, T + 1, (B,), generator=g).to(torch.int32)
    lens[0] = T
    if B > 1:
        lens[1] = 1
    return [t.to(cuda) for t in (x, w, b, gamma, beta, lens)]


def _ref(x, w, b, gamma, beta, lens, H, seed, keep):
    B, T, d = x.shape
    dh = d // H
    xf = x.float()
    qkv = (xf @ w.to(BF).float() + b).to(BF).float()          # the kernel rounds Q, K, V to bf16 (MFMA operands)
    q, k, v = (qkv[..., i * d:(i + 1) * d].view(B, T, H, dh).permute(0, 2, 1, 3) for i in range(3))
    sc = (q @ k.transpose(-1, -2)) / dh ** 0.5
    ar = torch.arange(T, device=x.device)
    km = (ar[None, :] < lens[:, None])[:, None, None, :]
    qm = (ar[None, :] < lens[:, None])[:, None, :, None]
    sc = torch.where(km, sc, torch.full_like(sc, PAD))
    p = torch.softmax(sc, dim=-1)
    p = torch.where(qm, p, torch.full_like(p, PAD))
    if 0.0 < keep < 1.0:
        m = O.dropout_mask(int(seed), B * H * T * T, keep).reshape(B, H, T, T)
        p = torch.where(torch.as_tensor(m, device=x.device), p / keep, torch.zeros_like(p))
    o = (p.to(BF).float() @ v).permute(0, 2, 1, 3).reshape(B, T, d)
    s = o + xf
    mean = s.mean(-1, keepdim=True)
    var = ((s - mean) ** 2).mean(-1, keepdim=True)
    return gamma * (s - mean) / torch.sqrt(var + 1e-8) + beta, s, qkv


@pytest.mark.parametrize("B,T", [(5, 50), (3, 64), (9, 33), (7, 32), (6, 17), (11, 10), (13, 1), (300, 50)])
@pytest.mark.parametrize("keep", [1.0, 0.9])
def test_mhsa_block_forward(cuda, B, T, keep):
    x, w, b, gamma, beta, lens = _mk(cuda, B, T, seed=B * 100 + T)
    img = torch.empty(ops.mhsa_image_bytes(), dtype=torch.uint8, device=cuda)
    ops.mhsa_image_build(w, img)
    seed = 0x1234567
    y, s, stats, qkv = ops.mhsa_block_fwd(x, lens, img, b, gamma, beta, 1e-8, 4, seed, keep)
    torch.cuda.synchronize()
    y_ref, s_ref, qkv_ref = _ref(x, w, b, gamma, beta, lens, 4, seed, keep)
    assert ((qkv.float() - qkv_ref).abs() / qkv_ref.abs().amax(-1, keepdim=True)).max() < 2.0 ** -7
    live = (torch.arange(T, device=cuda)[None, :] < lens[:, None])           # padded query rows hold -2^32-scale values: compare relatively
    tol = 2.0 ** -6
    es = ((s.float() - s_ref).abs() / s_ref.abs().amax(-1, keepdim=True).clamp_min(1e-3))
    assert es[live].max() < tol
    assert es[~live].max() < 3 * tol if (~live).any() else True
    ey = ((y.float() - y_ref).abs() / y_ref.abs().amax(-1, keepdim=True).clamp_min(1e-3))
    assert ey[live].max() < 2 * tol
    # statistics of the stored (rounded) s
    sm = s.float().mean(-1).reshape(-1)
    assert ((stats[:, 0] - sm).abs() / s.float().abs().amax(-1).reshape(-1).clamp_min(1e-3)).max() < 1e-3


def test_mhsa_block_statistics_of_large_mean_rows(cuda):
    """Rows whose mean is ~1000 x their spread (a one-pass E[s^2] - mean^2 in fp32 loses the variance to cancellation
    there, and with eps = 1e-8 the clamp at zero hands the LayerNorm gradient an rstd of 1e4).  The kernel accumulates the statistics
    shifted by the row's first input element; the saved (mean, rstd) must match the two-pass statistics of the stored s, which is
    what dmt_ln_fwd / dmt_ln_bwd of the three-launch path compute."""
    B, T = 7, 50
    x, w, b, gamma, beta, lens = _mk(cuda, B, T, seed=77)
    # bf16 spacing at 64 is 0.5: offsets of +-0.5, +-1 around 64 are exact in bf16 and survive the rounding of s
    g = torch.Generator().manual_seed(5)
    x = (64.0 + 0.5 * torch.randint(-2, 3, (B, T, 320), generator=g).float()).to(BF).to(cuda)
    w = w * 1e-3                                    # a small attention term: s stays x + O(1e-1)
    img = torch.empty(ops.mhsa_image_bytes(), dtype=torch.uint8, device=cuda)
    ops.mhsa_image_build(w, img)
    y, s, stats, _qkv = ops.mhsa_block_fwd(x, lens, img, b * 0.0, gamma, beta, 1e-8, 4, 0, 1.0)
    torch.cuda.synchronize()
    sf = s.float().reshape(-1, 320).double()
    mean = sf.mean(-1)
    var = ((sf - mean[:, None]) ** 2).mean(-1)
    rstd = 1.0 / torch.sqrt(var + 1e-8)
    live = (torch.arange(T, device=cuda)[None, :] < lens[:, None]).reshape(-1)
    assert (mean[live].abs() / var[live].sqrt()).min() > 50       # the case this test is about
    assert ((stats[:, 0].double() - mean).abs()[live] / mean[live].abs()).max() < 1e-6
    assert ((stats[:, 1].double() - rstd).abs()[live] / rstd[live]).max() < 1e-3
    yr = (gamma.double() * ((sf - mean[:, None]) * rstd[:, None]) + beta.double()).reshape(B, T, 320)
    assert ((y.double() - yr).abs()[live.reshape(B, T)]).max() < 2.0 ** -5


def test_mhsa_block_autograd_matches_unfused_path(cuda):
    B, T, d, H = 37, 50, 320, 4
    x, w, b, gamma, beta, lens = _mk(cuda, B, T, seed=77)
    img = torch.empty(ops.mhsa_image_bytes(), dtype=torch.uint8, device=cuda)
    ops.mhsa_image_build(w, img)
    wt = ops.Weight(w, w.to(BF), w.to(BF).t().contiguous())
    g = torch.Generator().manual_seed(5)
    dy = torch.randn(B, T, d, generator=g).to(BF).to(cuda)
    dy = dy * (torch.arange(T, device=cuda)[None, :, None] < lens[:, None, None])   # (padded rows carry no gradient downstream)
    outs = []
    for fused in (True, False):
        leaves = [t.clone().requires_grad_(True) for t in (w, b, gamma, beta)]
        xa = x.clone().requires_grad_(True)
        if fused:
            y = ops.MhsaBlockFn.apply(xa, leaves[0], leaves[1], wt, leaves[2], leaves[3], lens, H, img, 99, 0.9, 1e-8)
        else:
            s1 = ops.SelfAttnBlockFn.apply(xa, leaves[0], leaves[1], wt, lens, H, 99, 0.9)
            y = ops.layer_norm(s1, leaves[2], leaves[3])
        y.backward(dy)
        outs.append((y.detach().float(), xa.grad.float(), [l.grad.float() for l in leaves]))
    (ya, dxa, ga), (yb, dxb, gb) = outs
    live = (torch.arange(T, device=cuda)[None, :] < lens[:, None])
    assert (ya - yb)[live].abs().max() < 0.06
    assert (dxa - dxb).abs().max() / dxb.abs().max() < 3e-2
    for a, bb in zip(ga, gb):
        assert (a - bb).abs().max() / bb.abs().max() < 3e-2


@pytest.mark.parametrize("B,T", [(5, 50), (3, 64), (9, 33), (7, 32), (6, 17), (11, 10), (64, 50)])
@pytest.mark.parametrize("keep", [1.0, 0.9])
def test_mhsa_block_backward_matches_the_fp32_statement(cuda, B, T, keep):
    """Every gradient of the block (dx, dWqkv, dbias, dgamma, dbeta) against autograd through the plain PyTorch fp32 statement above (the
    bf16 roundings of Q | K | V and of the weights are straight-through there): LayerNorm gradient, attention gradient with the key mask
    before and the query mask after the softmax, the dropout mask the forward drew, dx = dqkv Wqkv^T + ds.  Tolerance: 3 % of the largest element of each gradient (bf16 ds, dqkv, P, dS)."""
    H, d = 4, 320
    x, w, b, gamma, beta, lens = _mk(cuda, B, T, seed=B * 100 + T + 1)
    img = torch.empty(ops.mhsa_image_bytes(), dtype=torch.uint8, device=cuda)
    ops.mhsa_image_build(w, img)
    wt = ops.Weight(w, w.to(BF), w.to(BF).t().contiguous())
    g = torch.Generator().manual_seed(11)
    live = (torch.arange(T, device=cuda)[None, :] < lens[:, None])
    dy = torch.randn(B, T, d, generator=g).to(BF).to(cuda) * live[:, :, None]           # (padded rows carry no gradient downstream)
    seed = 4242
    leaves = [t.clone().requires_grad_(True) for t in (w, b, gamma, beta)]
    xa = x.clone().requires_grad_(True)
    y = ops.MhsaBlockFn.apply(xa, leaves[0], leaves[1], wt, leaves[2], leaves[3], lens, H, img, seed, keep, 1e-8)
    y.backward(dy)
    torch.cuda.synchronize()
    ref_leaves = [t.clone().requires_grad_(True) for t in (w, b, gamma, beta)]
    xr = x.float().clone().requires_grad_(True)
    y_ref, _s, _qkv = _ref(xr, ref_leaves[0], ref_leaves[1], ref_leaves[2], ref_leaves[3], lens, H, seed, keep)
    # the padded query rows of the statement hold -2^32-scale values: they are cut out of the graph exactly as the model cuts them (no
    # gradient comes back into them)
    (y_ref * dy.float()).sum().backward()
    names = ("dx", "dWqkv", "dbias", "dgamma", "dbeta")
    got = [xa.grad.float()] + [l.grad.float() for l in leaves]
    want = [xr.grad] + [l.grad for l in ref_leaves]
    for nm, a, r in zip(names, got, want):
        if nm == "dx":
            a, r = a * live[:, :, None], r * live[:, :, None]
        assert torch.isfinite(a).all(), nm
        assert (a - r).abs().max() <= 3e-2 * r.abs().max() + 1e-6, (nm, float((a - r).abs().max()), float(r.abs().max()))


def _bwd_inputs(cuda, B, T, seed, ragged=True):
    H, d = 4, 320
    x, w, b, gamma, beta, lens = _mk(cuda, B, T, seed=seed)
    if not ragged:
        lens[:] = T
    img = torch.empty(ops.mhsa_image_bytes(), dtype=torch.uint8, device=cuda)
    ops.mhsa_image_build(w, img)
    imgb = torch.empty(ops.mhsa_bwd_image_bytes(), dtype=torch.uint8, device=cuda)
    ops.mhsa_bwd_image_build(w, imgb)
    g = torch.Generator().manual_seed(seed + 1)
    live = (torch.arange(T, device=cuda)[None, :] < lens[:, None])
    ds = torch.randn(B, T, d, generator=g).to(BF).to(cuda) * live[:, :, None]
    return x, w, b, gamma, beta, lens, img, imgb, ds, live


@pytest.mark.parametrize("B,T", [(5, 50), (4, 64), (3, 64), (9, 33), (8, 32), (6, 17), (16, 16), (11, 10), (13, 1), (300, 50)])
@pytest.mark.parametrize("keep", [1.0, 0.9])
def test_mhsa_block_bwd_kernel_matches_the_unfused_kernels(cuda, B, T, keep):
    """dmt_mhsa_block_bwd (attention gradient + dx = dqkv Wqkv^T + ds in one launch) against the two launches it replaces, on the same
    bf16 inputs: dmt_attn_bwd for dqkv, the [M, 960] x [960, 320] GEMM with its residual for dx.  Same masks, same dropout counter; both
    round P, dS and dqkv to bf16, so the two agree to a few bf16 roundings of the largest element -- and to nothing worse on any row."""
    H, d = 4, 320
    x, w, b, gamma, beta, lens, img, imgb, ds, live = _bwd_inputs(cuda, B, T, seed=B * 10 + T)
    y, s, stats, qkv = ops.mhsa_block_fwd(x, lens, img, b, gamma, beta, 1e-8, H, 4321, keep)
    dqkv, dx = ops.mhsa_block_bwd(ds, qkv, lens, imgb, H, 4321, keep)
    torch.cuda.synchronize()
    ref = torch.zeros_like(qkv)
    ops.attn_core_bwd(qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:], lens, lens, None, ds, ref[..., :d], ref[..., d:2 * d], ref[..., 2 * d:], H, 4321, keep)
    wt = ops.Weight(w, w.to(BF), w.to(BF).t().contiguous())
    dx_ref = ops.linear_backward_input(ref.view(B * T, 3 * d), wt, resid=ds.view(B * T, d)).view(B, T, d)
    torch.cuda.synchronize()
    assert torch.isfinite(dqkv.float()).all() and torch.isfinite(dx.float()).all()
    for name, lo in (("dq", 0), ("dk", d), ("dv", 2 * d)):
        a, r = dqkv[..., lo:lo + d].float(), ref[..., lo:lo + d].float()
        assert (a - r).abs().max() <= 2e-2 * r.abs().max() + 1e-6, (name, float((a - r).abs().max()), float(r.abs().max()))
    a, r = dx.float() * live[:, :, None], dx_ref.float() * live[:, :, None]
    assert (a - r).abs().max() <= 2e-2 * r.abs().max() + 1e-6, ("dx", float((a - r).abs().max()), float(r.abs().max()))


@pytest.mark.parametrize("B,T", [(5, 50), (3, 64), (9, 33), (7, 32), (6, 17), (11, 10), (64, 50)])
@pytest.mark.parametrize("keep", [1.0, 0.9])
def test_mhsa_block_with_the_fused_backward_matches_the_fp32_statement(cuda, B, T, keep):
    """MhsaBlockFn with image_bwd (LayerNorm gradient -> dmt_mhsa_block_bwd -> weight gradient): every gradient against autograd through the
    plain PyTorch fp32 statement, as test_mhsa_block_backward_matches_the_fp32_statement does for the four-launch backward."""
    H, d = 4, 320
    x, w, b, gamma, beta, lens, img, imgb, _ds, live = _bwd_inputs(cuda, B, T, seed=B * 100 + T + 1)
    wt = ops.Weight(w, w.to(BF), w.to(BF).t().contiguous())
    g = torch.Generator().manual_seed(11)
    dy = torch.randn(B, T, d, generator=g).to(BF).to(cuda) * live[:, :, None]
    seed = 4242
    leaves = [t.clone().requires_grad_(True) for t in (w, b, gamma, beta)]
    xa = x.clone().requires_grad_(True)
    with L_route() as rt:
        y = ops.MhsaBlockFn.apply(xa, leaves[0], leaves[1], wt, leaves[2], leaves[3], lens, H, img, seed, keep, 1e-8, None, imgb)
        y.backward(dy)
    torch.cuda.synchronize()
    assert rt.counts.get("dmt_mhsa_block_bwd", 0) == 1 and not any(k.startswith("dmt_attn_bwd") for k in rt.counts)
    ref_leaves = [t.clone().requires_grad_(True) for t in (w, b, gamma, beta)]
    xr = x.float().clone().requires_grad_(True)
    y_ref, _s, _qkv = _ref(xr, ref_leaves[0], ref_leaves[1], ref_leaves[2], ref_leaves[3], lens, H, seed, keep)
    (y_ref * dy.float()).sum().backward()
    names = ("dx", "dWqkv", "dbias", "dgamma", "dbeta")
    got = [xa.grad.float()] + [l.grad.float() for l in leaves]
    want = [xr.grad] + [l.grad for l in ref_leaves]
    for nm, a, r in zip(names, got, want):
        if nm == "dx":
            a, r = a * live[:, :, None], r * live[:, :, None]
        assert torch.isfinite(a).all(), nm
        assert (a - r).abs().max() <= 3e-2 * r.abs().max() + 1e-6, (nm, float((a - r).abs().max()), float(r.abs().max()))

"""Long-sequence attention core (64 < T <= 256; BASELINE configs[4]: clk / ord histories of 200, fp8 MFMA attention, batch 8192):
the flash-style kernels of dmt_attn_long.hip against the unfused form (batched GEMMs + dmt_softmax_*, independent code), the fp8
forward against the bf16 forward with its stated tolerance, the C-ABI's argument checks, and size-independent properties at full
size.  (tests/test_gpu_ops.py holds the comparisons with the fp64 oracle at T = 65, 130, 200.)"""
import ctypes as C

import numpy as np
import pytest
import torch

from cikm2020_dmt_amd import _lib as L
from cikm2020_dmt_amd import ops

pytestmark = pytest.mark.gpu


def _inputs(B, Tq, Tk, H, dh, cuda, seed=1):
    d = H * dh
    g = torch.Generator(device="cpu").manual_seed(seed)
    if Tq == Tk:
        pq = (torch.randn((B, Tq, 3 * d), generator=g) * 0.7).to(torch.bfloat16)
        pkv = None
    else:
        pq = (torch.randn((B, Tq, d), generator=g) * 0.7).to(torch.bfloat16)
        pkv = (torch.randn((B, Tk, 2 * d), generator=g) * 0.7).to(torch.bfloat16)
    x = torch.randn((B, Tq, d), generator=g).to(torch.bfloat16)
    rng = np.random.default_rng(seed)
    ql = torch.tensor(rng.integers(1, Tq + 1, size=B), dtype=torch.int32, device=cuda)
    kl = ql if pkv is None else torch.tensor(rng.integers(1, Tk + 1, size=B), dtype=torch.int32, device=cuda)
    w = torch.randn((B, Tq, d), generator=g).to(cuda) * (torch.arange(Tq, device=cuda)[None, :, None] < ql[:, None, None])
    return pq, pkv, x, ql, kl, w


def _run(pq, pkv, x, ql, kl, w, H, cuda, drop, fused=True, fp8=False):
    d = x.shape[2]
    a = pq.to(cuda).requires_grad_(True)
    bkv = pkv.to(cuda).requires_grad_(True) if pkv is not None else None
    xd = x.to(cuda).requires_grad_(True)
    out = ops.AttnFn.apply(a, bkv, xd, ql, kl, H, d, pkv is None, 0xC0FFEE if drop else 0, 0.9 if drop else 1.0,
                           ops.KernelOptions(attn_mma_fp8=fp8, attn_long_fused=fused))
    (out.float() * w).sum().backward()
    grads = [a.grad.float()] + ([bkv.grad.float()] if bkv is not None else []) + [xd.grad.float()]
    return out.detach().float(), grads


@pytest.mark.parametrize("drop", [False, True])
@pytest.mark.parametrize("B,Tq,Tk,H,dh", [(3, 65, 65, 2, 16), (2, 100, 100, 4, 80), (2, 128, 128, 2, 32), (3, 200, 200, 4, 80), (2, 208, 208, 2, 64),
                                          (2, 224, 224, 4, 80), (2, 256, 256, 2, 16), (3, 40, 200, 4, 80), (2, 200, 70, 4, 80), (4, 1, 200, 4, 80),
                                          (3, 1, 256, 2, 32), (2, 250, 90, 2, 64)])
def test_fused_long_kernels_match_the_unfused_form(cuda, B, Tq, Tk, H, dh, drop):
    """Forward and all input gradients, ragged query / key lengths, dropout on and off: both forms compute in bf16 with fp32
    accumulation from the same inputs and the same counter mask, so they agree to bf16 rounding of the intermediates
    (2e-2 of the largest value; a wrong mask, tile or dropout bit shows up as O(1))."""
    pq, pkv, x, ql, kl, w = _inputs(B, Tq, Tk, H, dh, cuda, seed=Tq + Tk)
    assert ops.long_fused_ok(H, *(t.to(cuda) for t in ([pq[..., :H * dh], pq[..., :H * dh]] if pkv is None else [pq, pkv[..., :H * dh]])))
    o1, g1 = _run(pq, pkv, x, ql, kl, w, H, cuda, drop, fused=True)
    o0, g0 = _run(pq, pkv, x, ql, kl, w, H, cuda, drop, fused=False)
    valid = (torch.arange(Tq, device=cuda)[None, :, None] < ql[:, None, None])
    eo = ((o1 - o0).abs() * valid).max().item() / (o0.abs() * valid).max().item()
    assert eo < 2e-2, eo
    # rows of padded queries carry -2^32+1 times a sum of V rows (reference behaviour): relative comparison
    pad = ~valid.expand_as(o0)
    if pad.any():
        ep = ((o1 - o0).abs()[pad] / (o0.abs()[pad] + 1.0)).max().item()
        assert ep < 3e-2, ep
    for a, b in zip(g1, g0):
        e = (a - b).abs().max().item() / (b.abs().max().item() + 1e-20)
        assert e < 4e-2, e


@pytest.mark.parametrize("B,T,H,dh", [(3, 200, 4, 80), (2, 128, 2, 64), (2, 70, 2, 16)])
def test_fp8_forward_stays_within_its_tolerance_of_the_bf16_kernel(cuda, B, T, H, dh):
    """mma_dtype = DMT_FP8_E4M3: Q, K, V and the weights are rounded to e4m3 (3 mantissa bits: relative step 2^-3 .. 2^-4 per element).
    Stated tolerance: the attention term (out - resid) within 10 % of its largest magnitude element-wise (worst of the 400-case sweep
    scripts/attn_long_fuzz.py: 9.7 %), and within 6 % in the RMS
    (measured 4.1 % at T = 200 with ragged lengths: an e4m3 product carries ~5 % rms error, and a peaked softmax averages over few of
    them); the backward pass is the bf16 one, unchanged."""
    pq, pkv, x, ql, kl, w = _inputs(B, T, T, H, dh, cuda, seed=9)
    x = torch.zeros_like(x)          # (no residual: the bf16 rounding of attn + x would otherwise dominate the comparison)
    o0, g0 = _run(pq, pkv, x, ql, kl, w, H, cuda, True, fp8=False)
    o8, g8 = _run(pq, pkv, x, ql, kl, w, H, cuda, True, fp8=True)
    valid = (torch.arange(T, device=cuda)[None, :, None] < ql[:, None, None])
    xa = x.to(cuda).float()
    a0, a8 = (o0 - xa) * valid, (o8 - xa) * valid
    assert (a8 - a0).abs().max().item() > 0                      # it IS a different arithmetic
    scale = a0.abs().max().item()
    assert (a8 - a0).abs().max().item() < 1e-1 * scale, ((a8 - a0).abs().max().item(), scale)
    rms = ((a8 - a0) ** 2).mean().sqrt().item() / (a0 ** 2).mean().sqrt().item()
    assert rms < 6e-2, rms
    pad = ~valid.expand_as(o0)
    if pad.any():
        # padded-query rows hold -2^32+1 times a (cancelling) sum of kept V rows: the constant rides in the output scale, V is rounded
        # to e4m3 -- error bounded by 2^32 / keep * sum_k |V[k]| * 2^-4
        d = H * dh
        bound = 2.0 ** 32 / 0.9 * pq[..., 2 * d:].float().abs().sum(dim=1, keepdim=True).to(cuda) * 2.0 ** -4
        assert bool(((o8 - o0).abs() <= bound.expand_as(o0))[pad].all())
    for a, b in zip(g8, g0):
        assert torch.equal(a, b)                                  # same bf16 backward kernel, same inputs


def test_long_entry_points_check_their_arguments(cuda):
    lib = L.load()
    assert lib.dmt_attn_long_supported(L.DMT_BF16, 80, 200, 200) == 1
    assert lib.dmt_attn_long_supported(L.DMT_BF16, 80, 257, 200) == 0
    assert lib.dmt_attn_long_supported(L.DMT_BF16, 20, 200, 200) == 0
    assert lib.dmt_attn_long_supported(L.DMT_F32, 80, 200, 200) == 0
    B, T, H, dh = 2, 100, 2, 16
    d = H * dh
    q = torch.randn((B, T, 3 * d), device=cuda).to(torch.bfloat16)
    out = torch.empty((B, T, d), dtype=torch.bfloat16, device=cuda)
    lens = torch.full((B,), T, dtype=torch.int32, device=cuda)
    desc = ops._attn_desc(torch.bfloat16, B, H, dh, T, T, q[..., :d], q[..., d:2 * d], q[..., 2 * d:], lens, lens, None, out)
    assert lib.dmt_attn_long_fwd(C.byref(desc), ops.stream_ptr()) == 0
    bad = ops._attn_desc(torch.bfloat16, B, H, dh, T, T, q[..., 1:d + 1], q[..., d:2 * d], q[..., 2 * d:], lens, lens, None, out)   # Q rows off by 2 bytes
    assert lib.dmt_attn_long_fwd(C.byref(bad), ops.stream_ptr()) == -1 and b"16-byte" in lib.dmt_last_error()
    bad2 = ops._attn_desc(torch.bfloat16, B, H, dh, T, T, q[..., :d], q[..., d:2 * d], q[..., 2 * d:], lens, lens, None, out)
    bad2.dh = 20
    assert lib.dmt_attn_long_fwd(C.byref(bad2), ops.stream_ptr()) == -3
    bad3 = ops._attn_desc(torch.bfloat16, B, H, dh, T, T, q[..., :d], q[..., d:2 * d], q[..., 2 * d:], lens, lens, None, out)
    bad3.mma_dtype = 7
    assert lib.dmt_attn_long_fwd(C.byref(bad3), ops.stream_ptr()) == -1
    torch.cuda.synchronize()


@pytest.mark.parametrize("fp8", [False, True])
def test_full_size_long_attention_properties(cuda, fp8):
    """BASELINE configs[4] at full size: batch 8192, L = 200, 4 heads of 80, ragged lengths, dropout off for the closed forms.
      (a) the weights of every valid query sum to 1: with V = 1 the attention term is exactly 1 (fp8: 1 is an e4m3 number, the
          weights' rounding errors remain: up to 2^-4 when one weight dominates);
      (b) examples are independent: rows of a 64-example slice recomputed alone are bit-identical;
      (c) the gradients of the slice (dropout ON, same counters as in the full batch need the same (b, h) -- so the slice is the
          head of the batch) are bit-identical too;
      (d) keys past k_len get no gradient; padded query rows send none to Q and K."""
    ko = ops.KernelOptions(attn_mma_fp8=fp8)
    B, T, H, dh = 8192, 200, 4, 80
    d = H * dh
    g = torch.Generator(device="cpu").manual_seed(3)
    qkv = (torch.randn((B, T, 3 * d), generator=g, dtype=torch.bfloat16) * 0.7).to(cuda)
    rng = np.random.default_rng(8)
    lens_np = rng.integers(1, T + 1, size=B)
    lens_np[:4] = [T, 1, 64, 65]
    lens = torch.tensor(lens_np, dtype=torch.int32, device=cuda)
    valid = (torch.arange(T, device=cuda)[None, :] < lens[:, None])
    # (a)
    ones = qkv.clone()
    ones[..., 2 * d:] = 1.0
    zero_resid = torch.zeros((B, T, d), dtype=torch.bfloat16, device=cuda)
    out = ops.AttnFn.apply(ones, None, zero_resid, lens, lens, H, d, True, 0, 1.0, ko).float()
    err = ((out - 1.0).abs() * valid[:, :, None]).max().item()
    assert err < (7e-2 if fp8 else 1e-2), err        # fp8: a dominant weight near 1 is rounded with relative error up to 2^-4
    del ones, out
    # (b), (c), (d)
    n = 64
    w = torch.randn((B, T, d), generator=g, dtype=torch.bfloat16).to(cuda) * valid[:, :, None]
    full = qkv.clone().requires_grad_(True)
    o_full = ops.AttnFn.apply(full, None, zero_resid, lens, lens, H, d, True, 0xBEEF, 0.9, ko)
    (o_full.float() * w).sum().backward()
    part = qkv[:n].clone().requires_grad_(True)
    o_part = ops.AttnFn.apply(part, None, zero_resid[:n], lens[:n], lens[:n], H, d, True, 0xBEEF, 0.9, ko)
    (o_part.float() * w[:n]).sum().backward()
    assert torch.equal(o_full[:n], o_part)
    assert torch.equal(full.grad[:n], part.grad)
    gk, gv, gq = full.grad[..., d:2 * d], full.grad[..., 2 * d:], full.grad[..., :d]
    assert float((gk.float().abs() * (~valid)[:, :, None]).max()) == 0.0
    assert float((gq.float().abs() * (~valid)[:, :, None]).max()) == 0.0
    assert torch.isfinite(full.grad.float()).all()
    assert float(gv.float().abs().max()) > 0 and float(gk.float().abs().max()) > 0

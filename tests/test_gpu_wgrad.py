"""dmt_wgrad320 (wide-block weight-gradient reduction) against a plain PyTorch fp32 matmul of the same bf16 operands.

Reference: the kernel / bias gradients of tf.layers.dense (TransformerModel_util.py:188-190, 224-228): dW = X^T dY, db = colsum(dY).
Tolerance: fp32 accumulation of exact bf16 products in a different order: |d| <= 2e-4 of the largest entry (M up to 200k terms).
"""
import pytest
import torch

from cikm2020_dmt_amd import ops

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


@pytest.mark.parametrize("M,N", [(31, 1280), (4096, 960), (20001, 1280), (65536 + 17, 960), (40000, 264)])
@pytest.mark.parametrize("transposed", [False, True])
def test_wgrad320_matches_fp32_matmul(cuda, M, N, transposed):
    g = torch.Generator().manual_seed(M + N)
    A = torch.randn(M, 320, generator=g).to(BF).to(cuda)
    Bm = torch.randn(M, N, generator=g).to(BF).to(cuda)
    ref = A.float().t() @ Bm.float()                       # [320, N]
    C0 = torch.randn(320, N, generator=g).to(cuda)
    if transposed:
        C = C0.t().contiguous().clone()                      # [N, 320]
        bias = torch.zeros(320, device=cuda)
        ops.wgrad320(A, Bm, C, True, bias, 2)
        got, bref = C.t() - C0, A.float().sum(0)
    else:
        C = C0.clone()
        bias = torch.zeros(N, device=cuda)
        ops.wgrad320(A, Bm, C, False, bias, 1)
        got, bref = C - C0, Bm.float().sum(0)
    torch.cuda.synchronize()
    scale = ref.abs().max()
    assert (got - ref).abs().max() / scale < 2e-4
    assert (bias - bref).abs().max() / bref.abs().max().clamp_min(1.0) < 2e-4


def test_wgrad320_with_padded_row_strides(cuda):
    g = torch.Generator().manual_seed(5)
    M = 30000
    Abig = torch.randn(M, 328, generator=g).to(BF).to(cuda)
    Bbig = torch.randn(M, 1000, generator=g).to(BF).to(cuda)
    A, Bm = Abig[:, :320], Bbig[:, 8:968]
    C = torch.zeros(320, 960, device=cuda)
    ops.wgrad320(A, Bm, C, False)
    ref = A.float().t() @ Bm.float()
    assert (C - ref).abs().max() / ref.abs().max() < 2e-4

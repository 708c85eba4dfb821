"""BASELINE-size checks (configs[1]: E64 dims, batch 4096, L = 50/50/10) through size-independent properties and sampled references.
(The whole full-size step also meets the oracle directly -- ~20 s of oracle time per form:
tests/test_gpu_parity_bf16.py::test_configs1_at_full_size_forward_loss_and_every_gradient_match_the_oracle; the checks here cover
single kernels at their full shapes and the properties that hold at any size.)

  * the persistent direct-to-LDS GEMMs with ~31 output tiles per workgroup (cross-tile prefetch, counted waits) against an fp32
    matmul of the same bf16 operands on a random sample of rows -- every epilogue the train step uses;
  * the split-K weight-gradient GEMM with its ones row over the full 204800-row reduction;
  * the whole train-step forward: permuting the examples permutes the logits BIT-EXACTLY (rows never interact);
  * the whole backward incl. the sparse embedding-gradient reduction: grad(full batch) = mean of the two half-batch gradients.
"""
import numpy as np
import pytest
import torch

from cikm2020_dmt_amd import ops
from cikm2020_dmt_amd import spec as S
from cikm2020_dmt_amd.data_feed.synthetic import make_batch
from cikm2020_dmt_amd.train import Trainer

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _sample_rows(M, n, seed):
    g = torch.Generator().manual_seed(seed)
    idx = torch.randint(0, M, (n,), generator=g)
    idx[0], idx[1], idx[2] = 0, M - 1, M // 2 + 127      # first / last row and a tile edge
    return idx


@pytest.mark.parametrize("N,K,epi", [(1280, 320, "bias_relu"), (320, 1280, "bias_resid"), (960, 320, "bias"), (320, 960, "plain"),
                                     (1280, 320, "gate"), (320, 1280, "gate_resid"), (640, 320, "bias")])
def test_forward_gemm_shapes_of_the_train_step(cuda, N, K, epi):
    M = 4096 * 50
    torch.manual_seed(N * 7 + K)
    x = (torch.randn(M, K, device=cuda) * 0.5).to(BF)
    wt = (torch.randn(N, K, device=cuda) * 0.05).to(BF)              # transposed bf16 shadow: both operands k-contiguous
    bias = torch.randn(N, device=cuda) if "bias" in epi else None
    resid = torch.randn(M, N, device=cuda).to(BF) if "resid" in epi else None
    gate = (torch.randn(M, N, device=cuda)).clamp_min(0).to(BF) if "gate" in epi else None
    out = torch.empty(M, N, dtype=BF, device=cuda)
    ops.gemm(x, K, 1, wt, 1, K, M, N, K, out, N, bias=bias, act_ncols=N if "relu" in epi else 0, gate=gate, ldg=N if gate is not None else 0,
             resid=resid, ldr=N if resid is not None else 0)
    rows = _sample_rows(M, 4096, 11).to(cuda)
    ref = x[rows].float() @ wt.float().t()
    if bias is not None:
        ref = ref + bias
    if "relu" in epi:
        ref = ref.clamp_min(0)
    if gate is not None:
        ref = ref * (gate[rows].float() > 0)
    if resid is not None:
        ref = ref + resid[rows].float()
    got = out[rows].float()
    scale = ref.abs().max().item()
    err = (got - ref).abs().max().item() / scale
    assert err < 6e-3, (epi, err)        # one bf16 rounding of the output: 2^-9 relative to the element, <= that of the max
    assert torch.isfinite(out).all()


@pytest.mark.parametrize("Kin,N", [(320, 1280), (1280, 320), (320, 960)])
def test_weight_gradient_gemm_full_reduction(cuda, Kin, N):
    R = 4096 * 50
    torch.manual_seed(Kin + N)
    x = (torch.randn(R, Kin, device=cuda) * 0.5).to(BF)
    dy = (torch.randn(R, N, device=cuda) * 0.1).to(BF)
    dW, db = ops.linear_backward_weight(x, dy, want_bias=True)
    ref_w = x.float().t() @ dy.float()
    ref_b = dy.float().sum(0)
    assert (dW - ref_w).abs().max().item() / ref_w.abs().max().item() < 2e-4       # fp32 accumulation, different order only
    assert (db - ref_b).abs().max().item() / ref_b.abs().max().item() < 2e-4


def _trainer(dropout=False):
    sp = S.e64_spec()
    return sp, Trainer(sp, device="cuda", compute_dtype=BF, seed=77, dropout=dropout)


def _subset(inputs, mask, idx):
    out = {}
    for k, v in inputs.items():
        if hasattr(v, "rows"):
            rows = v.rows()
            out[k] = type(v).from_rows([rows[i] for i in idx], dtype=np.asarray(v.values).dtype)
        else:
            out[k] = np.asarray(v)[idx]
    return out, mask[idx]


def test_full_size_forward_is_row_permutation_equivariant(cuda):
    sp, tr = _trainer()
    B = 4096
    inputs, mask, _ = make_batch(sp, B, seed=5, lengths="ragged")
    perm = np.random.default_rng(0).permutation(B)
    inputs_p, mask_p = _subset(inputs, mask, perm)
    b1 = tr.make_batch(inputs, mask)
    b2 = tr.make_batch(inputs_p, mask_p)
    tr.sync_rows(b1)
    (c1, o1), y1 = tr.engine.inference(b1)
    tr.sync_rows(b2)
    (c2, o2), y2 = tr.engine.inference(b2)
    p = torch.as_tensor(perm, device=cuda)
    assert torch.equal(c1.detach()[p], c2.detach()) and torch.equal(o1.detach()[p], o2.detach()) and torch.equal(y1.detach()[p], y2.detach())
    assert torch.isfinite(c1).all() and torch.isfinite(o1).all()


def test_full_size_gradient_is_mean_of_half_batch_gradients(cuda):
    sp, tr = _trainer()
    B = 4096
    inputs, mask, _ = make_batch(sp, B, seed=6, lengths="full")
    halves = [np.arange(0, B // 2), np.arange(B // 2, B)]

    def grads(inp, msk):
        b = tr.make_batch(inp, msk)
        tr.forward_backward(b)
        dense = tr.store.grads.clone()
        uniq, n_uniq, rows, _cap = tr.engine.sparse
        n = int(n_uniq.item())
        return dense, uniq[:n].clone().long(), rows[:n].clone()

    gd, ku, ru = grads(inputs, mask)
    parts = [grads(*_subset(inputs, mask, h)) for h in halves]
    # dense arena: mean of the halves
    dense_mean = 0.5 * (parts[0][0] + parts[1][0])
    scale = gd.abs().max().item()
    assert (gd - dense_mean).abs().max().item() / scale < 2e-2          # bf16 activations; weight gradients accumulate in fp32
    # sparse rows: scatter both halves into one map keyed by global row
    total = tr.store.total_rows
    D = ru.shape[1]
    acc = torch.zeros(total, D, device=cuda)
    for (_d, k, r) in parts:
        acc.index_add_(0, k, 0.5 * r)
    full = torch.zeros(total, D, device=cuda)
    full.index_add_(0, ku, ru)
    touched = torch.zeros(total, dtype=torch.bool, device=cuda); touched[ku] = True
    touched2 = torch.zeros(total, dtype=torch.bool, device=cuda); touched2[parts[0][1]] = True; touched2[parts[1][1]] = True
    assert torch.equal(touched, touched2)                                 # the same distinct rows
    s2 = full.abs().max().item()
    assert (full - acc).abs().max().item() / s2 < 2e-2


def test_full_size_dropout_on_uniform_ids_ragged_lengths(cuda):
    """BASELINE size with train-mode dropout ON (0.1 / 0.5), uniform ids (worst locality) and ragged lengths -- through
    the forward AND the backward.  Size-independent properties (the dropout counter of an element depends on its example's position
    in the batch only):
      * rows never interact: the first half's logits are bit-identical whether or not the second half is in the batch;
      * the loss is a mask-weighted mean over the batch (inference_mlp.py:207-213): with the second half's label rows zeroed, the
        gradient of the full batch is exactly half the gradient of the first half run alone -- dense arena and sparse rows."""
    sp, tr = _trainer(dropout=True)
    B = 4096
    inputs, mask, _ = make_batch(sp, B, seed=9, lengths="ragged", law="uniform")
    first = np.arange(0, B // 2)
    inputs_h, mask_h = _subset(inputs, mask, first)
    mask_a = mask.copy()
    mask_a[B // 2:] = 0.0

    def run(inp, msk):
        b = tr.make_batch(inp, msk)
        tr.forward_backward(b)
        (c, o), y = tr.last["out"]
        uniq, n_uniq, rows, _cap = tr.engine.sparse
        n = int(n_uniq.item())
        return (c.detach().clone(), o.detach().clone(), y.detach().clone()), tr.store.grads.clone(), uniq[:n].clone().long(), rows[:n].clone()

    (c1, o1, y1), gd, ku, ru = run(inputs, mask_a)
    (c2, o2, y2), gh, kh, rh = run(inputs_h, mask_h)
    tr.engine.dropout_step_seed = None
    (c0, _o0), _y0 = tr.engine.inference(tr.make_batch(inputs_h, mask_h))
    assert (c0.detach() - c2).abs().max().item() > 1e-3                      # dropout really was on in the runs above
    h = B // 2
    assert torch.equal(c1[:h], c2) and torch.equal(o1[:h], o2) and torch.equal(y1[:h], y2)
    assert torch.isfinite(gd).all() and torch.isfinite(ru).all()
    scale = gh.abs().max().item()
    assert (gd - 0.5 * gh).abs().max().item() / (0.5 * scale) < 2e-2
    total = tr.store.total_rows
    D = ru.shape[1]
    full = torch.zeros(total, D, device=cuda); full.index_add_(0, ku, ru)
    half = torch.zeros(total, D, device=cuda); half.index_add_(0, kh, 0.5 * rh)
    assert (full - half).abs().max().item() / half.abs().max().item() < 2e-2
    # the dropped fraction of the block input at this size: 0.1 within sampling noise (65 M elements)
    tr.engine.dropout_step_seed = 12345
    X, _tar, _z = tr.engine.gather(tr.make_batch(inputs, mask))
    x0 = X[0]
    if tr.engine._last_packs[0] is not None:
        # packed rows (engine.SeqPack: this ragged batch has ~50 % padding): every row of [1, R, d] is a live position
        assert x0.shape[1] == tr.engine._last_packs[0].R
        frac = (x0 == 0).float().mean().item()
    else:
        lens = tr.make_batch(inputs, mask).feats[sp["attention_embed_pairs"][0][-1][0]].lens
        live = (torch.arange(x0.shape[1], device=cuda)[None, :] < lens[:, None])
        frac = (x0[live] == 0).float().mean().item()
    assert abs(frac - 0.1) < 2e-3, frac

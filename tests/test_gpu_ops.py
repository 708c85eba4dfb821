"""Per-kernel parity against the oracle's reference functions (attention block, FFN, LayerNorm, loss, AUC, Adam)."""
import numpy as np
import pytest
import torch

from oracle import dmt_oracle as O
from oracle import dmt_oracle_torch as OT
from cikm2020_dmt_amd import ops
from cikm2020_dmt_amd import _lib as L

pytestmark = pytest.mark.gpu


def _rand(shape, seed, scale=1.0):
    return (np.random.default_rng(seed).standard_normal(shape) * scale)


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 5e-5), (torch.bfloat16, 3e-2)])
@pytest.mark.parametrize("B,T,H,dh", [(5, 50, 4, 20), (3, 10, 4, 20), (2, 64, 4, 80), (4, 7, 2, 16), (3, 1, 4, 20),
                                      # head dims of the coalesced MFMA kernels (dh % 16 == 0), ragged and full tiles
                                      (5, 50, 4, 80), (6, 10, 4, 16), (3, 33, 2, 32), (2, 50, 2, 64), (2, 64, 2, 16),
                                      # longer than the fused kernels' 64 keys: batched GEMMs + dmt_softmax_* (BASELINE long-seq variant)
                                      (3, 200, 4, 20), (2, 200, 4, 80), (2, 65, 2, 16), (2, 130, 2, 32)])
def test_attention_core_matches_oracle(cuda, dtype, tol, B, T, H, dh):
    d = H * dh
    q = torch.tensor(_rand((B, T, d), 1), dtype=dtype)
    k = torch.tensor(_rand((B, T, d), 2), dtype=dtype)
    v = torch.tensor(_rand((B, T, d), 3), dtype=dtype)
    x = torch.tensor(_rand((B, T, d), 4), dtype=dtype)
    lens = np.random.default_rng(5).integers(1, T + 1, size=B)
    qn, kn, vn, xn = (t.double().numpy() for t in (q, k, v, x))
    Q_ = np.concatenate(np.split(qn, H, axis=2), axis=0)
    K_ = np.concatenate(np.split(kn, H, axis=2), axis=0)
    V_ = np.concatenate(np.split(vn, H, axis=2), axis=0)
    m = O.sequence_mask(lens, T)
    ref = np.concatenate(np.split(O.scaled_dot_product_attention(Q_, K_, V_, m, m), H, axis=0), axis=2) + xn
    qkv = torch.cat([q, k, v], dim=-1).to(cuda).requires_grad_(True)
    xd = x.to(cuda).requires_grad_(True)
    ld = torch.tensor(lens, dtype=torch.int32, device=cuda)
    out = ops.AttnFn.apply(qkv, None, xd, ld, ld, H, d, True)
    got = out.detach().double().cpu().numpy()
    valid = m[:, :, None]
    # rows of padded queries hold c * sum(V) with c = -2**32+1 (reference behaviour): compare relatively
    err = np.abs(got - ref) / (np.abs(ref) + 1.0)
    # (padded query rows sum T terms of magnitude 4e9 * |V| with cancellation: their fp32 rounding error grows with T)
    assert err.max() < (tol if T <= 64 or dtype != torch.float32 else 4 * tol), err.max()
    # gradients (only through valid query rows, as in the model) vs torch autograd of the second oracle
    w = torch.tensor(_rand((B, T, d), 6) * m[:, :, None], dtype=torch.float64)
    (out.double() * w.to(cuda)).sum().backward()
    qt, kt, vt, xt = (torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (qn, kn, vn, xn))
    Qh = qt.view(B, T, H, dh).transpose(1, 2); Kh = kt.view(B, T, H, dh).transpose(1, 2); Vh = vt.view(B, T, H, dh).transpose(1, 2)
    S = (Qh @ Kh.transpose(-1, -2)) / dh ** 0.5
    km = torch.tensor(m)[:, None, None, :]
    S = torch.where(km, S, torch.full_like(S, OT.PADDING_NUM))
    A = torch.softmax(S, -1)
    A = torch.where(torch.tensor(m)[:, None, :, None], A, torch.full_like(A, OT.PADDING_NUM))
    Oref = (A @ Vh).transpose(1, 2).reshape(B, T, d) + xt
    (Oref * w).sum().backward()
    gq = qkv.grad.double().cpu()
    for name, g, r in (("dq", gq[..., :d], qt.grad), ("dk", gq[..., d:2 * d], kt.grad), ("dv", gq[..., 2 * d:], vt.grad),
                       ("dx", xd.grad.double().cpu(), xt.grad)):
        e = (g - r).abs().max().item() / (r.abs().max().item() + 1e-12)
        assert e < 10 * tol, (name, e, r.abs().max().item())


@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-5), (torch.bfloat16, 2e-2)])
@pytest.mark.parametrize("rows,d", [(37, 80), (1000, 320), (5, 7), (64, 1024)])
def test_layernorm_matches_oracle(cuda, dtype, tol, rows, d):
    x = torch.tensor(_rand((rows, d), 1, 3.0), dtype=dtype)
    g = torch.tensor(1.0 + 0.1 * _rand((d,), 2), dtype=torch.float32)
    b = torch.tensor(0.1 * _rand((d,), 3), dtype=torch.float32)
    ref = O.ln(x.double().numpy(), g.double().numpy(), b.double().numpy())
    xd = x.to(cuda).requires_grad_(True)
    gd = g.to(cuda).requires_grad_(True)
    bd = b.to(cuda).requires_grad_(True)
    y = ops.layer_norm(xd, gd, bd, 1e-8)
    assert np.abs(y.detach().double().cpu().numpy() - ref).max() < tol * 5
    w = torch.tensor(_rand((rows, d), 4), dtype=torch.float64)
    (y.double() * w.to(cuda)).sum().backward()
    xt = x.double().clone().requires_grad_(True); gt = g.double().clone().requires_grad_(True); bt = b.double().clone().requires_grad_(True)
    (OT._ln(xt, gt, bt) * w).sum().backward()
    for name, a, r in (("dx", xd.grad, xt.grad), ("dgamma", gd.grad, gt.grad), ("dbeta", bd.grad, bt.grad)):
        e = (a.double().cpu() - r).abs().max().item() / (r.abs().max().item() + 1e-12)
        assert e < 20 * tol, (name, e)


def test_loss_kernel_matches_oracle(cuda):
    rng = np.random.default_rng(0)
    B = 333
    c = rng.standard_normal(B) * 3; o = rng.standard_normal(B) * 3 - 2; yb = rng.standard_normal(B)
    c[:4] = [40.0, -40.0, 17.0, -17.0]      # saturate the clip branches of the keras cross entropy
    cls = rng.integers(0, 5, size=B)
    mask = np.zeros((B, 5), np.float32); mask[np.arange(B), cls] = 1
    so = O.default_spec()
    for method in ("two_head_add", "two_head_multiply"):
        for rel in ("ctr_rel", "ctr"):
            ref = O.loss_multi_task_unbias(((c.reshape(-1, 1), o.reshape(-1, 1)), yb.reshape(-1, 1)), mask, so, method, rel)
            ct, ot, yt = (torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (c, o, yb))
            lt = OT.loss_unbias(((ct.reshape(-1, 1), ot.reshape(-1, 1)), yt.reshape(-1, 1)), mask, so, method, rel)
            lt.backward()
            cd, od, yd = (torch.tensor(a, dtype=torch.float32, device=cuda, requires_grad=True) for a in (c, o, yb))
            wc = torch.tensor(so["weight_ctr"], dtype=torch.float32, device=cuda)
            wo = torch.tensor(so["weight_ecvr"], dtype=torch.float32, device=cuda)
            loss, pc, pv = ops.LossUnbiasFn.apply(cd, od, yd, torch.tensor(mask, device=cuda), wc, wo, so["loss_weight"],
                                                  1 if method == "two_head_multiply" else 0, 1 if rel == "ctr_rel" else 0)
            loss.backward()
            assert abs(float(loss) - ref) / abs(ref) < 2e-5, (method, rel)
            for a, r in ((cd.grad, ct.grad), (od.grad, ot.grad), (yd.grad, yt.grad)):
                assert (a.double().cpu() - r).abs().max().item() < 2e-5 * max(1.0, r.abs().max().item())


def test_auc_histogram_matches_tf_metrics_auc(cuda):
    rng = np.random.default_rng(1)
    B = 5000
    pred = rng.random(B).astype(np.float32)
    pred[:50] = np.round(pred[:50] * 199) / 199          # exactly on thresholds
    label = (rng.random(B) < 0.2 + 0.5 * pred).astype(np.float32)
    hist = torch.zeros(2 * 201, dtype=torch.int64, device=cuda)
    pred_d, label_d = torch.tensor(pred, device=cuda), torch.tensor(label, device=cuda)
    L.call("dmt_auc_hist", B, ops.p(pred_d), ops.p(label_d), 200, ops.p(hist), ops.stream_ptr())
    from cikm2020_dmt_amd.metrics import auc_from_hist
    got = auc_from_hist(hist.cpu().numpy(), 200)
    ref = O.tf_metrics_auc(label, pred, 200)
    h = hist.cpu().numpy().reshape(2, 201)
    thr = np.array([-1e-7] + [(i + 1) / 199.0 for i in range(198)] + [1.0 + 1e-7]).astype(np.float32)
    ref_bins = (pred[:, None] > thr[None, :]).sum(1)
    ref_h = np.zeros((2, 201), np.int64)
    np.add.at(ref_h, (label.astype(np.int64), ref_bins), 1)
    assert np.array_equal(h, ref_h), np.nonzero(h != ref_h)
    assert abs(got - ref) < 1e-9, (got, ref)


def test_lazy_adam_is_bitwise_equal_to_dense_sweep(cuda):
    """Size-independent property: replaying zero-gradient steps lazily == sweeping every row every step."""
    from cikm2020_dmt_amd.variables import VariableStore
    from cikm2020_dmt_amd.optim import TFAdam
    from tests.util import small_specs
    _so, sp = small_specs()
    a = VariableStore(sp, cuda, torch.float32, seed=1)
    b = VariableStore(sp, cuda, torch.float32, seed=1)
    oa, ob = TFAdam(a), TFAdam(b)
    rng = np.random.default_rng(0)
    D = max(t.shape[1] for t in a.table.values())
    for step in range(12):
        n = int(rng.integers(1, 400))
        rows = np.unique(rng.integers(0, a.total_rows, size=n)).astype(np.int32)
        if step % 3 == 0:
            rows = rows[rows % 7 == 0] if (rows % 7 == 0).any() else rows[:1]      # leave long gaps for most rows
        g = rng.standard_normal((len(rows), D)).astype(np.float32) * 0.01
        uniq = torch.tensor(rows, device=cuda)
        gr = torch.tensor(g, device=cuda)
        nu = torch.tensor([len(rows)], dtype=torch.int32, device=cuda)
        oa.begin(); oa.apply_sparse((uniq, nu, gr, len(rows))); oa.end()
        dense = {}
        for name, (base, nr) in b.table_rows.items():
            dim = b.tables[name].shape[1]
            gd = np.zeros((nr, dim), np.float32)
            sel = (rows >= base) & (rows < base + nr)
            gd[rows[sel] - base] = g[sel][:, :dim]
            dense[name] = torch.tensor(gd, device=cuda)
        ob.begin(); ob.apply_dense_tables(dense); ob.end()
    oa.flush_tables()
    ob_p, oa_p = b.tab_p.cpu().numpy(), a.tab_p.cpu().numpy()
    assert np.array_equal(oa_p.view(np.uint32), ob_p.view(np.uint32))
    assert np.array_equal(a.tab_m.cpu().numpy().view(np.uint32), b.tab_m.cpu().numpy().view(np.uint32))
    assert np.array_equal(a.tab_v.cpu().numpy().view(np.uint32), b.tab_v.cpu().numpy().view(np.uint32))


def test_rows_reduce_bf16_matches_fp32_reduce_of_rounded_rows(cuda):
    """DP merge in bf16 mode: out[seg[e]] += bf16 in_rows[vals[e]] with fp32 accumulation."""
    import ctypes as C
    from cikm2020_dmt_amd import _lib as L, ops
    torch.manual_seed(3)
    dev = torch.device("cuda")
    N, D, R = 1000, 24, 37
    keys = torch.randint(0, R, (N,), dtype=torch.int32, device=dev)
    keys[::17] = R                                   # invalid entries (padding of shorter ranks)
    order = torch.argsort(keys.to(torch.int64), stable=True)
    skeys = keys[order].contiguous()
    svals = order.to(torch.int32).contiguous()
    uniq, inv = torch.unique(skeys.to(torch.int64), return_inverse=True)
    seg = inv.to(torch.int32).contiguous()
    rows = torch.randn(N, D, device=dev)
    rows_lp = rows.to(torch.bfloat16).contiguous()
    out = torch.zeros(len(uniq), D, device=dev)
    L.call("dmt_rows_reduce_bf16", ops.p(skeys), ops.p(svals), ops.p(seg), N, R, ops.p(rows_lp), ops.p(out), D, None, 0, ops.stream_ptr())
    ref = torch.zeros(len(uniq), D, device=dev)
    valid = skeys < R
    ref.index_add_(0, inv[valid], rows_lp.float()[order][valid])
    n_valid_rows = int((uniq < R).sum())
    assert torch.allclose(out[:n_valid_rows], ref[:n_valid_rows], atol=1e-5, rtol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols,dtype", [(512, 50 * 320, torch.bfloat16), (300, 10 * 24, torch.bfloat16), (64, 50 * 24, torch.float32)])
def test_colsum_drop_equals_column_sum_of_dropped_gradient(cuda, rows, cols, dtype):
    """Learned-position gradient under block-input dropout (TransformerModel.py:101 + TransformerModel_util.py:296-306):
    sum_b (dX * mask / keep)[b, :] with the counter mask dmt_dropout applies for the same seed -- both the 16-byte and the
    scalar kernel."""
    g = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn((rows, cols), generator=g).to(dtype).to(cuda)
    seed, keep = 0x1234567, 0.9
    dropped = torch.empty_like(x)
    L.call("dmt_dropout", ops.dt_code(dtype), x.numel(), ops.p(x), ops.p(dropped), seed, keep, ops.stream_ptr())
    mask = (dropped != 0) | (x == 0)
    want = (x.double() * mask / keep).sum(0)
    out = torch.zeros(cols, dtype=torch.float32, device=cuda)
    L.call("dmt_colsum_drop", ops.dt_code(dtype), rows, cols, ops.p(x), 1.0, ops.p(out), seed, keep, 0, ops.stream_ptr())
    torch.cuda.synchronize()
    assert 0.85 < mask.float().mean().item() < 0.95
    np.testing.assert_allclose(out.cpu().numpy(), want.cpu().numpy(), rtol=2e-5, atol=2e-4)


@pytest.mark.gpu
def test_batched_cast_transpose_equals_per_weight_casts(cuda):
    """One launch for every 2-D weight (ragged shapes, padded transposed strides) == the per-weight kernel, bit for bit."""
    g = torch.Generator(device="cpu").manual_seed(9)
    shapes = [(3047, 72), (320, 960), (33, 31), (1, 5), (64, 64)]
    triples, want = [], []
    for i, (k, n) in enumerate(shapes):
        src = torch.randn((k, n), generator=g).to(cuda)
        kpad = (k + 7) // 8 * 8
        dp = torch.zeros((k, n), dtype=torch.bfloat16, device=cuda) if i != 2 else None
        dt = torch.zeros((n, kpad), dtype=torch.bfloat16, device=cuda)[:, :k] if i != 3 else None
        triples.append((src, dp, dt))
        want.append(src.to(torch.bfloat16))
    table = ops.cast_shadow_jobs(triples, cuda)
    ops.cast_shadow_batched(table)
    torch.cuda.synchronize()
    for (src, dp, dt), w in zip(triples, want):
        if dp is not None:
            assert torch.equal(dp, w)
        if dt is not None:
            assert torch.equal(dt, w.t())


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,H,dh", [(7, 50, 4, 80), (9, 10, 4, 80), (3, 64, 2, 16), (4, 33, 2, 64), (3, 200, 4, 80), (2, 96, 2, 16)])
def test_mfma_attention_dropout_matches_fp32_kernel(cuda, B, T, H, dh):
    """Attention-weight dropout (TransformerModel_util.py:51 tf.layers.dropout on the softmax) in the coalesced bf16 MFMA kernels
    against the scalar fp32 kernel of the same library (independent code, same counter mask): a different mask would show up as
    O(1) errors.  Ragged lengths, forward and all three input gradients."""
    d = H * dh
    g = torch.Generator(device="cpu").manual_seed(11)
    qkv32 = (torch.randn((B, T, 3 * d), generator=g) * 0.7)
    x32 = torch.randn((B, T, d), generator=g)
    lens = torch.tensor(np.random.default_rng(3).integers(1, T + 1, size=B), dtype=torch.int32, device=cuda)
    w = torch.randn((B, T, d), generator=g).to(cuda) * (torch.arange(T, device=cuda)[None, :, None] < lens[:, None, None])
    outs, grads = [], []
    for dt in (torch.float32, torch.bfloat16):
        qkv = qkv32.to(torch.bfloat16).to(dt).to(cuda).requires_grad_(True)      # same (bf16-representable) inputs on both paths
        x = x32.to(torch.bfloat16).to(dt).to(cuda).requires_grad_(True)
        out = ops.AttnFn.apply(qkv, None, x, lens, lens, H, d, True, 0x5EED1234, 0.9)
        (out.float() * w).sum().backward()
        outs.append(out.detach().float())
        grads.append(qkv.grad.detach().float())
    valid = (torch.arange(T, device=cuda)[None, :, None] < lens[:, None, None])
    eo = ((outs[0] - outs[1]).abs() * valid).max().item() / outs[0].abs().mul(valid).max().item()
    eg = (grads[0] - grads[1]).abs().max().item() / grads[0].abs().max().item()
    assert eo < 2e-2, eo
    assert eg < 4e-2, eg


@pytest.mark.gpu
@pytest.mark.parametrize("B,Tq,Tk,H,dh", [(5, 12, 50, 4, 80), (4, 40, 20, 2, 64), (3, 9, 9, 2, 16), (4, 1, 200, 4, 80), (3, 30, 100, 2, 16)])
def test_mfma_cross_attention_matches_fp32_kernel(cuda, B, Tq, Tk, H, dh):
    """Cross attention with different query / key tile counts (coalesced kernels <NTQ, NTK> = <1,2>, <2,1>, <1,1>), separate
    query and key lengths, packed K|V: bf16 MFMA path against the scalar fp32 kernel, forward and gradients."""
    d = H * dh
    g = torch.Generator(device="cpu").manual_seed(17)
    q32, kv32, x32 = torch.randn((B, Tq, d), generator=g) * 0.7, torch.randn((B, Tk, 2 * d), generator=g) * 0.7, torch.randn((B, Tq, d), generator=g)
    rng = np.random.default_rng(4)
    ql = torch.tensor(rng.integers(1, Tq + 1, size=B), dtype=torch.int32, device=cuda)
    kl = torch.tensor(rng.integers(1, Tk + 1, size=B), dtype=torch.int32, device=cuda)
    valid = (torch.arange(Tq, device=cuda)[None, :, None] < ql[:, None, None])
    w = torch.randn((B, Tq, d), generator=g).to(cuda) * valid
    outs, gq, gkv = [], [], []
    for dt in (torch.float32, torch.bfloat16):
        q = q32.to(torch.bfloat16).to(dt).to(cuda).requires_grad_(True)
        kv = kv32.to(torch.bfloat16).to(dt).to(cuda).requires_grad_(True)
        x = x32.to(torch.bfloat16).to(dt).to(cuda)
        out = ops.AttnFn.apply(q, kv, x, ql, kl, H, d, False, 0xABCDEF, 0.9)
        (out.float() * w).sum().backward()
        outs.append(out.detach().float()); gq.append(q.grad.detach().float()); gkv.append(kv.grad.detach().float())
    eo = ((outs[0] - outs[1]).abs() * valid).max().item() / outs[0].abs().mul(valid).max().item()
    assert eo < 2e-2, eo
    for a, b2 in ((gq[0], gq[1]), (gkv[0], gkv[1])):
        e = (a - b2).abs().max().item() / a.abs().max().item()
        assert e < 4e-2, e


@pytest.mark.gpu
def test_sparse_adam_bf16_rows_and_padding_keys(cuda):
    """dmt_adam_sparse_rows_bf16 (reduced rows straight off the data-parallel wire) == dmt_adam_sparse_rows on the same rows
    widened to fp32, bit for bit; keys >= the total row count (padding slots of a rank-major gathered list) are skipped by
    both."""
    from cikm2020_dmt_amd.variables import VariableStore
    from cikm2020_dmt_amd.optim import TFAdam
    from tests.util import small_specs
    _so, sp = small_specs()
    a = VariableStore(sp, cuda, torch.float32, seed=2)
    b = VariableStore(sp, cuda, torch.float32, seed=2)
    oa, ob = TFAdam(a), TFAdam(b)
    rng = np.random.default_rng(1)
    D = max(t.shape[1] for t in a.table.values())
    D = (D + 3) // 4 * 4
    for step in range(5):
        rows = np.unique(rng.integers(0, a.total_rows, size=300)).astype(np.int32)
        pad = np.full(17, a.total_rows, dtype=np.int32)                      # padding slots, scattered through the list
        keys = np.concatenate([rows[:100], pad[:9], rows[100:], pad[9:]])
        g = (rng.standard_normal((len(keys), D)) * 0.01).astype(np.float32)
        gb = torch.tensor(g, device=cuda).to(torch.bfloat16)
        uniq = torch.tensor(keys, device=cuda)
        nu = torch.tensor([len(keys)], dtype=torch.int32, device=cuda)
        oa.begin(); oa.apply_sparse((uniq, nu, gb, len(keys)), grad_scale=0.5); oa.end()
        ob.begin(); ob.apply_sparse((uniq, nu, gb.float(), len(keys)), grad_scale=0.5); ob.end()
    for x, y in ((a.tab_p, b.tab_p), (a.tab_m, b.tab_m), (a.tab_v, b.tab_v)):
        assert torch.equal(x, y)
    assert torch.equal(a.last_step, b.last_step)
    assert int((a.tab_m != 0).sum()) > 0

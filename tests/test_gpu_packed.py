"""Packed rows (include/dmt_hip.h "PACKED ROWS", engine.SeqPack): a behaviour sequence computed on its real rows only.

The reference computes every sequence at [B, T_max] and masks (TransformerModel_util.py:43-48, 90-97; SURVEY.md F13: "a new kernel may
skip padded rows entirely").  Rows past an example's length are never read as keys and receive zero gradient, so the packed and the
dense layout must agree on everything that exists:
  * forward: logits of the same weights on the same ragged batch, packed against dense, to bf16 rounding of one accumulation order
    (the fused block tiles short examples with a smaller padded length: the same products, possibly summed in another order);
  * every gradient, packed against dense (weight gradients are sums over rows: the order of the sum changes, nothing else);
  * train-mode dropout draws the same masks (the counters keep the dense element index);
  * full-length batches are left in the dense layout (nothing to skip).
The oracle comparison of the packed path itself is tests/test_gpu_e64.py (the "packed" cases).
"""
import numpy as np
import pytest
import torch

from cikm2020_dmt_amd import _lib as L
from cikm2020_dmt_amd import ops
from cikm2020_dmt_amd import spec as S
from cikm2020_dmt_amd.data_feed.synthetic import make_batch
from cikm2020_dmt_amd.engine import SeqPack
from cikm2020_dmt_amd.train import Trainer
from oracle import dmt_oracle as O
from tests.test_gpu_e64 import E64_ROWS, _params
from tests.util import sparse_to_dense_tables


def test_seqpack_layout_on_the_host():
    """Rows disjoint and dense in [0, R); every example in exactly one block entry of its length class; entry fields consistent."""
    rng = np.random.default_rng(0)
    for B, T in ((100, 50), (1, 50), (37, 10), (300, 64)):
        lens = rng.integers(1, T + 1, size=B).astype(np.int32)
        lens[0] = T
        p = SeqPack(lens, T, torch.device("cpu"))
        assert p.R == int(lens.sum()) and sum(p.n_cls) == B
        ro, order = p.row_off.numpy(), p.order.numpy()
        cover = np.zeros(p.R, int)
        for b in range(B):
            cover[ro[b]: ro[b] + lens[b]] += 1
        assert (cover == 1).all() and sorted(order.tolist()) == list(range(B))
        cls = np.where(lens <= 16, 0, np.where(lens <= 32, 1, 2))
        assert (np.diff(cls[order]) >= 0).all() and p.n_short == int((cls < 2).sum())
        blk = p.blocks.view(p.n_tiles, 8, 2, 4).numpy()
        seen = []
        for t in range(p.n_tiles):
            lg = int(blk[t, 0, 0, 3])
            assert (blk[t, :, :, 3] == lg).all()
            for rb in range(8):
                if lg >= 5:
                    assert (blk[t, rb, 0] == blk[t, rb, 1]).all()
                if lg == 6 and rb % 2 == 1:
                    assert (blk[t, rb] == blk[t, rb - 1]).all()
                for h in range(2 if lg == 4 else 1):
                    ex, ln, off, _ = blk[t, rb, h]
                    if ex >= 0 and not (lg == 6 and rb % 2 == 1):
                        assert ln == lens[ex] and off == ro[ex] and cls[ex] == lg - 4
                        seen.append(int(ex))
        assert sorted(seen) == list(range(B))
    assert not SeqPack.eligible(np.array([3, 0, 2], np.int32), 10) and not SeqPack.eligible(np.array([3, 11], np.int32), 10)
    assert SeqPack.eligible(np.array([3, 10], np.int32), 10) and not SeqPack.eligible(None, 10)


def _pair(cuda, B, seed, dropout, lengths="ragged"):
    sp = S.scaled_spec(S.e64_spec(), E64_ROWS)
    P = _params(dict(sp))
    inputs, mask, label = make_batch(sp, B, seed=seed, lengths=lengths, weights="random")
    out = []
    for packed in (False, True):
        tr = Trainer(sp, device=cuda, compute_dtype=torch.bfloat16, init=False, dropout=dropout, dropout_seed=77, packed_rows=packed)
        tr.store.load_state(P)
        out.append((tr, tr.make_batch(inputs, mask, label)))
    return sp, out


@pytest.mark.gpu
@pytest.mark.parametrize("B,dropout", [(40, False), (40, True), (700, True)])
def test_packed_and_dense_steps_agree(cuda, monkeypatch, B, dropout):
    monkeypatch.setattr(ops, "WGRAD320_MIN_ROWS", 1024)
    sp, ((td, bd), (tp, bp)) = _pair(cuda, B, seed=9, dropout=dropout)
    res = []
    for tr, batch in ((td, bd), (tp, bp)):
        with L.route_trace() as rt:
            loss = float(tr.forward_backward(batch))
            torch.cuda.synchronize()
        (c, o), yb = tr.last["out"]
        g = dict(tr.store.grad_dict())
        g.update(sparse_to_dense_tables(tr.store, tr.engine.sparse))
        res.append((loss, [x.detach().float().cpu().numpy() for x in (c, o, yb)], g, dict(rt.counts)))
    (ld, od, gd, rd), (lp, op_, gp, rp) = res
    assert rd.get("dmt_mhsa_block_fwd", 0) == 3 and rd.get("dmt_mhsa_block_fwd(packed)", 0) == 0
    assert rp.get("dmt_mhsa_block_fwd(packed)", 0) == 3 and rp.get("dmt_mhsa_block_fwd", 0) == 0 and rp.get("dmt_colsum_rows_packed", 0) == 3
    packs = [tp.engine.intermediates["pack_%d" % i] for i in range(3)]
    assert all(pk is not None and pk.R < 0.9 * B * pk.T for pk in packs)
    dl = max(np.abs(a - b).max() for a, b in zip(od, op_))
    print("packed vs dense, B = %d, dropout %s: max |dlogit| %.3g, loss %.6f / %.6f" % (B, dropout, dl, ld, lp))
    assert dl < 4e-3 and abs(ld - lp) < 2e-3 * abs(ld)          # (logits are O(1); one bf16 ulp at 1 is 7.8e-3)
    gscale = max(np.abs(v).max() for v in gd.values())
    worst = []
    for name, ref in gd.items():
        e = float(np.linalg.norm(gp[name] - ref) / max(np.linalg.norm(ref), 1e-3 * gscale * np.sqrt(ref.size)))
        worst.append((e, name))
    worst.sort(reverse=True)
    print("worst relative gradient distances:", worst[:4])
    assert worst[0][0] < 2e-2, worst[:6]


@pytest.mark.gpu
def test_packed_rows_train_like_dense_rows(cuda, monkeypatch):
    """Three Adam steps on the same three ragged batches: the parameters end where the dense run's do (to the rounding above)."""
    monkeypatch.setattr(ops, "WGRAD320_MIN_ROWS", 1024)
    sp = S.scaled_spec(S.e64_spec(), E64_ROWS)
    P = _params(dict(sp))
    batches = [make_batch(sp, 48, seed=300 + i, lengths="ragged", weights="random") for i in range(3)]
    ends = []
    for packed in (False, True):
        tr = Trainer(sp, device=cuda, compute_dtype=torch.bfloat16, init=False, dropout=True, dropout_seed=5, packed_rows=packed)
        tr.store.load_state(P)
        losses = [float(tr.train_step(tr.make_batch(*b))) for b in batches]
        tr.opt.flush_tables()
        torch.cuda.synchronize()
        ends.append((losses, {k: v.copy() for k, v in tr.store.state_dict().items()}))
    (l0, s0), (l1, s1) = ends
    print("losses dense / packed:", l0, l1)
    assert max(abs(a - b) for a, b in zip(l0, l1)) < 5e-3 * max(l0)
    # Adam moves every weight by ~lr per step in the direction of sign(m): an element whose gradient is rounding-sized may flip, so single
    # elements differ by up to the three steps' travel (3 lr = 3e-3); the BULK of every tensor must sit where the dense run put it
    # (the K bias of an attention block has NO gradient in exact arithmetic -- it shifts all scores of a softmax alike --: what Adam sees
    #  there is rounding noise, and sign(noise) * lr is what it applies: those tensors random-walk in either run)
    for k in s0:
        diff = np.abs(s1[k] - s0[k])
        assert diff.max() <= 3.2e-3, (k, float(diff.max()))
        if not k.endswith("attention/dense_1/bias"):
            assert diff.mean() <= 2.5e-4, (k, float(diff.mean()))


@pytest.mark.gpu
def test_full_length_batches_stay_dense_and_inference_takes_packed_rows_too(cuda):
    sp, ((td, bd), (tp, bp)) = _pair(cuda, 16, seed=3, dropout=False, lengths="full")
    with L.route_trace() as rt:
        tp.engine.inference(bp)
        torch.cuda.synchronize()
    assert rt.counts.get("dmt_mhsa_block_fwd", 0) == 3 and rt.counts.get("dmt_mhsa_block_fwd(packed)", 0) == 0
    sp, ((td, bd), (tp, bp)) = _pair(cuda, 64, seed=4, dropout=False)
    with torch.no_grad():
        (c0, o0), y0 = td.engine.inference(bd)
        with L.route_trace() as rt:
            (c1, o1), y1 = tp.engine.inference(bp)
            torch.cuda.synchronize()
    assert rt.counts.get("dmt_mhsa_block_fwd(packed)", 0) == 3
    assert max((a.float() - b.float()).abs().max().item() for a, b in ((c0, c1), (o0, o1), (y0, y1))) < 4e-3


@pytest.mark.gpu
def test_packed_position_gradient_kernel_against_torch(cuda):
    """dmt_colsum_rows_packed alone: dP[t] = sum over the examples that have row t, with and without the dropout mask, atomics and ordered."""
    rng = np.random.default_rng(1)
    B, T, d = 300, 50, 320
    lens = rng.integers(1, T + 1, size=B).astype(np.int32)
    pk = SeqPack(lens, T, torch.device(cuda))
    x = torch.randn((pk.R, d), device=cuda).to(torch.bfloat16)
    lens_d = torch.tensor(lens, device=cuda)
    ro = pk.row_off.cpu().numpy()
    dense = torch.zeros((B, T, d), dtype=torch.float32, device=cuda)
    for b in range(B):
        dense[b, : lens[b]] = x[ro[b]: ro[b] + lens[b]].float()
    for keep, seed in ((1.0, 0), (0.9, 0xBEEF)):
        ref = dense.clone()
        if keep < 1.0:
            m = torch.as_tensor(O.dropout_mask(seed, B * T * d, keep).reshape(B, T, d), device=cuda)
            ref = torch.where(m, ref / keep, torch.zeros_like(ref))
        ref = ref.sum(0)
        for ordered in (0, 1):
            out = torch.zeros((T, d), dtype=torch.float32, device=cuda)
            L.call("dmt_colsum_rows_packed", L.DMT_BF16, B, T, d, ops.p(x), ops.p(pk.row_off), ops.p(lens_d), 1.0, ops.p(out), seed, keep, ordered,
                   ops.stream_ptr())
            torch.cuda.synchronize()
            assert (out - ref).abs().max().item() < 2e-3 * ref.abs().max().item()


@pytest.mark.gpu
@pytest.mark.parametrize("packed,B,dropout", [(False, 40, True), (True, 40, True), (True, 700, True), (False, 300, False)])
def test_fused_self_attention_backward_in_the_engine(cuda, monkeypatch, packed, B, dropout):
    """DMTEngine.use_mhsa_bwd (DMT_FUSED_MHSA_BWD=1): the block's backward as LayerNorm gradient -> dmt_mhsa_block_bwd -> weight gradient,
    in the dense layout and on packed rows (all three padded lengths 16 / 32 / 64 occur in a ragged batch), against the four-launch
    backward of the same engine settings: same loss (the forward is the same), every gradient to the bf16 rounding of dS / dqkv."""
    monkeypatch.setattr(ops, "WGRAD320_MIN_ROWS", 1024)
    sp = S.scaled_spec(S.e64_spec(), E64_ROWS)
    P = _params(dict(sp))
    inputs, mask, label = make_batch(sp, B, seed=19, lengths="ragged", weights="random")
    res = []
    for fused in (False, True):
        tr = Trainer(sp, device=cuda, compute_dtype=torch.bfloat16, init=False, dropout=dropout, dropout_seed=77, packed_rows=packed)
        tr.engine.use_mhsa_bwd = fused
        tr.store.load_state(P)
        batch = tr.make_batch(inputs, mask, label)
        with L.route_trace() as rt:
            loss = float(tr.forward_backward(batch))
            torch.cuda.synchronize()
        g = dict(tr.store.grad_dict())
        g.update(sparse_to_dense_tables(tr.store, tr.engine.sparse))
        res.append((loss, g, dict(rt.counts)))
        tr.close()
    (l0, g0, r0), (l1, g1, r1) = res
    key = "dmt_mhsa_block_bwd(packed)" if packed else "dmt_mhsa_block_bwd"
    assert r1.get(key, 0) == 3 and r0.get(key, 0) == 0
    assert not any(k.startswith("dmt_attn_bwd(mfma") for k in r1)          # (the decoders' one-query attention has its own routes)
    assert l0 == l1
    gscale = max(np.abs(v).max() for v in g0.values())
    worst = []
    for name, ref in g0.items():
        e = float(np.linalg.norm(g1[name] - ref) / max(np.linalg.norm(ref), 1e-3 * gscale * np.sqrt(ref.size)))
        worst.append((e, name))
    worst.sort(reverse=True)
    print("fused vs four-launch backward (packed %s, B %d): worst relative gradient distances:" % (packed, B), worst[:4])
    assert worst[0][0] < 2e-2, worst[:6]

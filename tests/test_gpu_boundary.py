"""The reference-shaped boundary: Inference.inference(is_train) dropout semantics, online_inference, l2_norm,
streaming precision / recall, and the optimizer's long-run behaviour (lr-history restart, bounded lazy replay)."""
import numpy as np
import pytest
import torch

from oracle import dmt_oracle as O
from cikm2020_dmt_amd.data_feed.synthetic import make_batch
from cikm2020_dmt_amd.metrics import StreamingPrecisionRecall, precision_recall_reference
from cikm2020_dmt_amd.model.inference_mlp import Inference
from cikm2020_dmt_amd.optim import TFAdam
from cikm2020_dmt_amd.train import Trainer
from cikm2020_dmt_amd.variables import VariableStore
from tests.util import small_specs, sparse_to_dense_tables

pytestmark = pytest.mark.gpu


def _inference(cuda, dtype=torch.float32, seed=9):
    so, sp = small_specs()
    so = dict(so, dropout_rate=0.1, dropout_rate_bias=[0.5, 0.5])
    P = O.init_params(so, seed=seed)
    inf = Inference(None, device=cuda, compute_dtype=dtype, seed=4, spec=sp)
    inf.rt.store.load_state(P)
    return so, sp, P, inf


def test_facade_is_train_applies_the_reference_dropout(cuda):
    """inference(inputs, is_train=True) runs Transformer dropout 0.1 + bias-tower dropout 0.5 (TransformerModel.py:101,151;
    mmoe_transformer_unbias.py:274-278); is_train=False and is_predict never drop.  Same counter mask as the oracle."""
    so, sp, P, inf = _inference(cuda)
    inputs, mask, _label = make_batch(sp, 19, seed=2, lengths="ragged", weights="random")
    (c0, o0), yb0 = O.inference(inputs, P, so)
    with torch.no_grad():
        (c, o), yb = inf.inference(inputs, is_train=False)
    assert np.abs(c.cpu().numpy() - c0).max() < 3e-4 and np.abs(yb.cpu().numpy() - yb0).max() < 3e-4
    inf.set_step_seed(4242)
    with torch.no_grad():
        (cd, od), ybd = inf.inference(inputs, is_train=True)
    (c1, o1), yb1 = O.inference(inputs, P, so, step_seed=4242)
    assert np.abs(c1 - c0).max() > 1e-3                                   # dropout changes the outputs ...
    assert np.abs(cd.cpu().numpy() - c1).max() < 3e-4                     # ... exactly as the oracle's masks do
    assert np.abs(od.cpu().numpy() - o1).max() < 3e-4
    assert np.abs(ybd.cpu().numpy() - yb1).max() < 3e-4
    # the running counter: two training calls use different masks; predict resets to no dropout
    with torch.no_grad():
        (ca, _), _ = inf.inference(inputs, is_train=True)
        (cb, _), _ = inf.inference(inputs, is_train=True)
        cp, _op = inf.inference(inputs, is_train=True, is_predict=True)
    assert (ca - cb).abs().max() > 1e-4
    assert np.abs(cp.cpu().numpy() - c0).max() < 3e-4


def test_online_inference_equals_the_tiled_predict_graph(cuda):
    """inference_mlp.py:73-143: user-side id lists arrive once (1-D) and are tiled across BatchSize candidates."""
    so, sp, P, inf = _inference(cuda)
    B = 23
    inputs, _mask, _label = make_batch(sp, B, seed=6, lengths="ragged", weights="ones")
    from tests.test_gpu_serving import _tile_user_side
    tiled = _tile_user_side(sp, inputs)
    c_or, o_or = O.inference(tiled, P, so, is_predict=True)
    req = {"BatchSize": B, "features": inputs["features"]}
    for (_n, _r, _d, f, side) in list(sp["embedding_list"]) + list(sp["embedding_list_bias"]):
        spv, w = inputs[f], inputs[f + "Wts"]
        if side == "u":
            sel = np.asarray(spv.indices)[:, 0] == 0
            req[f] = np.asarray(spv.values)[sel]
            req[f + "Wts"] = np.asarray(w.values)[sel]
        else:
            req[f], req[f + "Wts"] = spv, w
    c, o = inf.online_inference(req)
    assert np.abs(c.cpu().numpy() - c_or).max() < 3e-4
    assert np.abs(o.cpu().numpy() - o_or).max() < 3e-4


def test_l2_norm_matches_the_reference_rule(cuda):
    """mmoe_transformer_unbias.py:42-60: sum_f l2_loss(E_f[unique ids of f]) * l2_emb_lambda / batch_size."""
    so, sp, P, inf = _inference(cuda)
    B = 31
    inputs, _m, _l = make_batch(sp, B, seed=3, lengths="ragged", weights="ones")
    want = 0.0
    for (name, _rows, _dim, feat, _side) in sp["embedding_list"]:
        ids = np.unique(np.asarray(inputs[feat].values).astype(np.int64))
        E = P["embedding_trans/%s/embedding" % name].astype(np.float64)
        want += 0.5 * (E[ids] ** 2).sum()
    want *= 0.01 / B                                # Inference(None): l2_emb_lambda 0.01 (dmt.conf:76), batch_size = this batch
    got = float(inf.l2_norm(inputs).detach())
    assert abs(got - want) / want < 1e-5


def test_l2_norm_gradient_joins_the_sparse_rows(cuda):
    """loss + l2_norm (run_dnn.py:174-175, wnd_wd > 1e-5): the embedding-gradient rows of one backward pass with the term minus those
    without it = lambda / batch_size * E[row] per embedding_list entry that holds the row (oracle/dmt_oracle.py:l2_norm)."""
    so, sp, P, inf = _inference(cuda)
    B = 23
    inputs, mask, label = make_batch(sp, B, seed=5, lengths="ragged", weights="ones")
    store = inf.rt.store
    dense = []
    for with_l2 in (False, True):
        store.zero_grad()
        logits = inf.inference(inputs, is_train=False)
        loss = inf.loss_multi_task_unbias(logits, label, mask, False, "two_head_add", "ctr")
        if with_l2:
            l2 = inf.l2_norm(inputs)
            loss = loss + 3.0 * l2                    # (a factor on the way: the row term must carry dLoss/dl2)
        loss.backward()
        dense.append(sparse_to_dense_tables(store, inf.rt.engine.sparse))
    want_val, want = O.l2_norm(inputs, P, so, 0.01, B)
    assert abs(float(l2.detach()) - want_val) <= 1e-5 * want_val
    seen = 0
    for name, g0 in dense[0].items():
        diff = dense[1][name] - g0
        ref = 3.0 * want.get(name, np.zeros_like(diff))
        # (the difference of two fp32 row sums of size ~1e-2, accumulated by atomics in either run: ~1e-8 of rounding)
        assert np.abs(diff - ref).max() <= 1e-4 * np.abs(ref).max() + 2e-8, name
        seen += int(np.abs(ref).max() > 0)
    assert seen == len(want) and seen >= 3
    # a table two entries read gets both terms
    entries = {}
    for (name, _r, _d, _f, _s) in sp["embedding_list"]:
        entries[name] = entries.get(name, 0) + 1
    assert max(entries.values()) >= 2


def test_trainer_adds_l2_norm_when_wnd_wd_is_set(cuda):
    """run_dnn.py:174-175 through the Trainer (wnd_wd > 1e-5): loss and embedding-gradient rows differ from the plain step's by the
    oracle's l2 term; a whole train_step (sparse rows finished on the index lane) runs with it."""
    so, sp = small_specs()
    P = O.init_params(so, seed=9)
    B = 17
    inputs, mask, label = make_batch(sp, B, seed=6, lengths="ragged", weights="ones")
    res = []
    for wd in (0.0, 1e-3):
        tr = Trainer(sp, device=cuda, compute_dtype=torch.float32, init=False, dropout=False, wnd_wd=wd, l2_emb_lambda=0.5)
        tr.store.load_state(P)
        loss = float(tr.forward_backward(tr.make_batch(inputs, mask, label)))
        res.append((loss, sparse_to_dense_tables(tr.store, tr.engine.sparse), tr))
    val, want = O.l2_norm(inputs, P, so, 0.5, B)
    assert abs((res[1][0] - res[0][0]) - val) <= 1e-4 * val
    for name, g0 in res[0][1].items():
        ref = want.get(name, np.zeros_like(g0))
        assert np.abs(res[1][1][name] - g0 - ref).max() <= 1e-4 * max(np.abs(ref).max(), 1e-9) + 2e-8, name
    tr = res[1][2]
    l1 = float(tr.train_step(tr.make_batch(inputs, mask, label)))
    l2 = float(tr.train_step(tr.make_batch(inputs, mask, label)))
    assert abs(l1 - res[1][0]) <= 1e-5 * l1 and l2 < l1


def test_l2_norm_rows_survive_the_sparse_lane(cuda):
    """wnd_wd > 1e-5 with the id-bound tail of backward deferred to the index lane (DMT_SPARSE_LANE=1): the l2 row term must land in the
    rows finish_sparse_backward() produces -- in EVERY step, not only the first (from step 2 on the `sparse` getter used to add
    it to the previous step's buffer, which was then zeroed).  Three steps, tables and dense parameters against the plain schedule."""
    from cikm2020_dmt_amd import ops
    so, sp = small_specs()
    P = O.init_params(so, seed=9)
    B = 33
    ops.set_deterministic(True)
    try:
        states = []
        for lane, wd in ((False, 1e-3), (True, 1e-3), (True, 0.0)):
            tr = Trainer(sp, device=cuda, compute_dtype=torch.float32, init=False, dropout=False, wnd_wd=wd, l2_emb_lambda=0.5)
            tr.store.load_state(P)
            tr.sparse_lane = lane
            for s in range(3):
                inputs, mask, label = make_batch(sp, B, seed=60 + s, lengths="ragged", weights="ones")
                tr.train_step(tr.make_batch(inputs, mask, label))
                assert tr.engine._pending_sparse is None
            tr.opt.flush_tables()
            torch.cuda.synchronize()
            states.append(tr.store.state_dict())
    finally:
        ops.set_deterministic(False)
    moved = 0
    for k in states[0]:
        assert np.array_equal(states[0][k], states[1][k]), k            # same rows, same order of sums: bit for bit
        moved += int(not np.array_equal(states[1][k], states[2][k]))
    assert moved > 0                                                     # (and the term is not a no-op in this setting)


def test_streaming_precision_recall_matches_the_tf_metrics_rule(cuda):
    rng = np.random.default_rng(0)
    m = StreamingPrecisionRecall(cuda)
    assert m.result() == (0.0, 0.0)                 # safe division before any data
    scores, labels = [], []
    for n in (1, 257, 4096, 33):
        s = rng.random(n).astype(np.float32)
        s[::7] = 0.5                                # exactly at the threshold: tf.greater(p, 0.5) is False
        y = (rng.random(n) < 0.3).astype(np.float32) * rng.integers(1, 3, n)     # labels are sums of mask columns: any non-zero is positive
        scores.append(s); labels.append(y)
        m.update(torch.as_tensor(s).to(cuda), torch.as_tensor(y.astype(np.float32)).to(cuda))
    p, r = m.result()
    pw, rw = precision_recall_reference(scores, labels)
    assert abs(p - pw) < 1e-12 and abs(r - rw) < 1e-12
    tp, fp, fn, tn = m.counts.cpu().numpy()
    assert tp + fp + fn + tn == sum(len(s) for s in scores)


def test_optimizer_survives_a_restored_large_step_and_a_full_lr_history(cuda):
    """restoring model.ckpt-1500000 (reset_slots(1500000)) must not trip the lr-history capacity, and a run longer than
    the capacity restarts the history (flush + rebase) instead of raising."""
    _so, sp = small_specs()
    st = VariableStore(sp, cuda, torch.float32, seed=1)
    opt = TFAdam(st, max_steps=8)
    opt.reset_slots(1500000)
    rng = np.random.default_rng(1)
    D = max(t.shape[1] for t in st.table.values())
    ref = VariableStore(sp, cuda, torch.float32, seed=1)
    oref = TFAdam(ref, max_steps=1 << 12)
    oref.reset_slots(1500000)
    for step in range(30):                          # 30 steps through a history of 8 entries: several rebases
        rows = np.unique(rng.integers(0, st.total_rows, size=50)).astype(np.int32)
        g = (rng.standard_normal((len(rows), D)) * 0.01).astype(np.float32)
        for o in (opt, oref):
            o.begin()
            o.apply_sparse((torch.tensor(rows, device=cuda), torch.tensor([len(rows)], dtype=torch.int32, device=cuda), torch.tensor(g, device=cuda), len(rows)))
            o.end()
    assert opt.global_step == 1500030
    opt.flush_tables(); oref.flush_tables()
    assert np.array_equal(st.tab_p.cpu().numpy().view(np.uint32), ref.tab_p.cpu().numpy().view(np.uint32))
    assert np.array_equal(st.tab_m.cpu().numpy().view(np.uint32), ref.tab_m.cpu().numpy().view(np.uint32))


def test_lazy_adam_long_gaps_are_bounded_and_match_the_dense_sweep(cuda):
    """A row untouched for thousands of steps: p exactly what the dense sweep leaves (it stops moving after ~150 zero-gradient
    steps), m and v within 1e-4 relative of the dense sweep's serial products (closed-form tail, dmt_optim.hip)."""
    _so, sp = small_specs()
    a = VariableStore(sp, cuda, torch.float32, seed=2)
    b = VariableStore(sp, cuda, torch.float32, seed=2)
    oa, ob = TFAdam(a, max_steps=1 << 13), TFAdam(b, max_steps=1 << 13)
    rng = np.random.default_rng(4)
    D = max(t.shape[1] for t in a.table.values())
    rows = np.unique(rng.integers(0, a.total_rows, size=300)).astype(np.int32)
    g = (rng.standard_normal((len(rows), D)) * 0.02).astype(np.float32)

    def dense_grads(rows_, g_):
        out = {}
        for name, (base, nr) in b.table_rows.items():
            dim = b.tables[name].shape[1]
            gd = np.zeros((nr, dim), np.float32)
            if rows_ is not None:
                sel = (rows_ >= base) & (rows_ < base + nr)
                gd[rows_[sel] - base] = g_[sel][:, :dim]
            out[name] = torch.tensor(gd, device=cuda)
        return out

    def sparse(rows_, g_):
        return (torch.tensor(rows_, device=cuda), torch.tensor([len(rows_)], dtype=torch.int32, device=cuda), torch.tensor(g_, device=cuda), len(rows_))

    oa.begin(); oa.apply_sparse(sparse(rows, g)); oa.end()
    ob.begin(); ob.apply_dense_tables(dense_grads(rows, g)); ob.end()
    zero = dense_grads(None, None)
    one = np.array([int(rows[0])], dtype=np.int32)
    for _ in range(3000):                           # 3000 steps in which only one row gets gradients
        g1 = (rng.standard_normal((1, D)) * 0.02).astype(np.float32)
        oa.begin(); oa.apply_sparse(sparse(one, g1)); oa.end()
        d = {k: v for k, v in zero.items()}
        for name, (base, nr) in b.table_rows.items():
            if base <= one[0] < base + nr:
                gd = torch.zeros_like(zero[name]); gd[one[0] - base] = torch.tensor(g1[0, : gd.shape[1]], device=cuda); d[name] = gd
        ob.begin(); ob.apply_dense_tables(d); ob.end()
    oa.flush_tables()
    pa, pb = a.tab_p.cpu().numpy(), b.tab_p.cpu().numpy()
    assert np.array_equal(pa.view(np.uint32), pb.view(np.uint32))
    for x, y in ((a.tab_m, b.tab_m), (a.tab_v, b.tab_v)):
        xa, ya = x.cpu().numpy().astype(np.float64), y.cpu().numpy().astype(np.float64)
        big = np.abs(ya) > 1e-30
        assert (np.abs(xa - ya)[big] / np.abs(ya)[big]).max() < 1e-4
        assert np.abs(xa[~big]).max() < 1e-29


def test_two_trainers_with_different_kernel_options_in_one_process(cuda):
    """Kernel choices are fields of an engine (ops.KernelOptions), not process-wide switches: a Trainer(attn_dtype="fp8") and a
    Trainer(attn_dtype="bf16") alive in the same process, used alternately, each take their own long-sequence forward kernel; so do
    two engines that differ in the streamed-weight projection."""
    from cikm2020_dmt_amd import _lib as L
    from cikm2020_dmt_amd import spec as S
    sp = dict(S.scaled_spec(S.e64_spec(), {"Sku": 2000, "Brand": 300, "Shopid": 300, "Cid3": 120}), maxlen_k=200)
    long_feats = {grp[0][0]: 200 for grp in sp["attention_embed_pairs"][:2]}
    inputs, mask, _ = make_batch(sp, 8, seed=3, lengths="full", seq_lens=long_feats)
    t8 = Trainer(sp, device=cuda, compute_dtype=torch.bfloat16, seed=1, dropout=False, attn_dtype="fp8")
    tb = Trainer(sp, device=cuda, compute_dtype=torch.bfloat16, seed=1, dropout=False, attn_dtype="bf16")
    tn = Trainer(sp, device=cuda, compute_dtype=torch.bfloat16, seed=1, dropout=False)
    tn.engine.kopts = tn.engine.kopts.replace(use_proj=False)
    assert t8.engine.kopts.attn_mma_fp8 and not tb.engine.kopts.attn_mma_fp8
    outs = {}
    for rnd in range(2):
        for name, tr in (("fp8", t8), ("bf16", tb), ("noproj", tn)):
            with L.route_trace() as rt:
                loss = float(tr.forward_backward(tr.make_batch(inputs, mask)))
                torch.cuda.synchronize()
            f8, fb, pj = (rt.counts.get(k, 0) for k in ("dmt_attn_long_fwd(fp8)", "dmt_attn_long_fwd", "dmt_proj"))
            assert (f8 > 0 and fb == 0) if name == "fp8" else (f8 == 0 and fb > 0), (name, rt.counts)
            assert (pj == 0) if name == "noproj" else (pj > 0), (name, rt.counts)
            outs.setdefault(name, []).append(loss)
    for name, v in outs.items():
        assert abs(v[0] - v[1]) < 1e-3 * abs(v[0]), (name, v)        # the same step twice (atomics: not bitwise)
    assert outs["fp8"][0] != outs["bf16"][0]                          # another arithmetic
    assert abs(outs["fp8"][0] - outs["bf16"][0]) < 5e-2 * abs(outs["bf16"][0])


def test_two_trainers_deferring_differently_in_one_process(cuda):
    """The state of a step in flight (collected long-row weight gradients, their lane, the long-row threshold) belongs to the engine
    (ops.StepState), not to the module: trainer A counts every weight gradient of >= 64 rows as long-row (dmt_wgrad320 runs), trainer B
    keeps the default threshold (no dmt_wgrad320 at this batch); used alternately in one process each takes its own route, and each
    ends where it ends when it runs alone."""
    from cikm2020_dmt_amd import _lib as L
    from cikm2020_dmt_amd import ops
    from cikm2020_dmt_amd import spec as S
    sp = S.scaled_spec(S.e64_spec(), {"Sku": 2000, "Brand": 300, "Shopid": 300, "Cid3": 120})
    batches = [make_batch(sp, 8, seed=40 + i, lengths="ragged") for i in range(3)]
    ops.set_deterministic(False)

    def fresh(min_rows):
        return Trainer(sp, device=cuda, compute_dtype=torch.bfloat16, seed=7, dropout=False, wgrad320_min_rows=min_rows)

    def run_alone(min_rows):
        tr = fresh(min_rows)
        return [float(tr.train_step(tr.make_batch(i, m))) for (i, m, _l) in batches]

    alone_a, alone_b = run_alone(64), run_alone(None)
    ta, tb = fresh(64), fresh(None)
    assert ta.engine.step_state is not tb.engine.step_state
    got_a, got_b = [], []
    for (i, m, _l) in batches:
        with L.route_trace() as rt:
            got_a.append(float(ta.train_step(ta.make_batch(i, m))))
            torch.cuda.synchronize()
        assert rt.counts.get("dmt_wgrad320", 0) > 0, sorted(rt.counts)
        with L.route_trace() as rt:
            got_b.append(float(tb.train_step(tb.make_batch(i, m))))
            torch.cuda.synchronize()
        assert rt.counts.get("dmt_wgrad320", 0) == 0, sorted(rt.counts)
        assert ops.deferred_wgrads_pending() == 0
    # The wide-block weight gradients (trainer a) add their row chunks with fp32 atomics: two runs of the SAME trainer agree on every
    # gradient to an ulp (1.5e-7 relative, measured) -- and Adam turns an ulp into a step where a gradient cancels to rounding noise
    # (update = lr g / (|g| + eps): elements with |g| ~ 1e-9 move by a different fraction of lr, or the other way).  Sixteen runs of
    # trainer a alone on these three batches: the first loss identical, the second within 2e-4 relative, the third within 6e-3 (eight
    # examples per batch); trainer b (no atomics at these sizes) repeats bit for bit.  So: a within that spread, b tightly.
    assert got_a[0] == alone_a[0] and np.allclose(got_a, alone_a, rtol=1.5e-2), (got_a, alone_a)
    assert abs(got_a[1] - alone_a[1]) <= 5e-4 * alone_a[1], (got_a, alone_a)
    assert np.allclose(got_b, alone_b, rtol=2e-3), (got_b, alone_b)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_early_catch_up_of_the_next_batch_changes_nothing(cuda, monkeypatch, dtype):
    """train_step(batch, prefetch=next): the pending zero-gradient Adam updates of the next batch's rows are replayed on the index lane
    while the current step runs (Trainer._catch_up_early) instead of in front of the next gather.  Same arithmetic, applied earlier:
    in deterministic mode six steps over batches with disjoint and overlapping ids end in bit-identical variables, Adam slots and
    last-touch steps, with and without it; and with it the catch-up in front of the gather finds nothing left to replay."""
    from cikm2020_dmt_amd import _lib as L
    from cikm2020_dmt_amd import ops
    so, sp = small_specs()
    P = O.init_params(so, seed=3)
    batches = [make_batch(sp, 9, seed=60 + (i % 4), lengths="ragged", weights="random") for i in range(7)]      # ids recur with gaps of 1..4 steps
    res = []
    ops.set_deterministic(True)
    try:
        for early in (True, False):
            tr = Trainer(sp, device=cuda, compute_dtype=dtype, init=False, dropout=True, dropout_seed=5)
            tr.early_catchup = early
            tr.store.load_state(P)
            bs = [tr.make_batch(i, m) for (i, m, _l) in batches]
            losses = []
            for s in range(6):
                losses.append(float(tr.train_step(bs[s], prefetch=bs[s + 1])))
            assert (tr.opt.stamp is not None) == early
            last = tr.store.last_step.clone()
            tr.opt.flush_tables()                  # every row through the last step: what a dense sweep holds
            torch.cuda.synchronize()
            res.append((losses, tr.store.tab_p.clone(), tr.store.tab_m.clone(), tr.store.tab_v.clone(), last, tr.store.params.clone()))
    finally:
        ops.set_deterministic(False)
    a, b = res
    assert a[0] == b[0]
    for x, y in zip(a[1:4], b[1:4]):
        assert torch.equal(x, y)
    assert torch.equal(a[5], b[5])
    # the early pass leaves rows of the prefetched batch at a LATER last-touch step than the lazy form (they are already up to date)
    assert bool((a[4] >= b[4]).all()) and bool((a[4] > b[4]).any())


def test_a_step_that_raises_after_its_early_begin_is_rolled_back(cuda):
    """Round-3 advice: with the early catch-up on, train_step begins the optimizer step (device step counter, lr history, row stamps)
    BEFORE forward / backward.  A step that then raises must leave the optimizer as if it had never begun (TFAdam.abort_step): the
    retried step and everything after it end bit-identical to a run in which nothing failed."""
    from cikm2020_dmt_amd import ops
    so, sp = small_specs()
    P = O.init_params(so, seed=3)
    batches = [make_batch(sp, 9, seed=60 + (i % 4), lengths="ragged", weights="random") for i in range(6)]
    res = []
    ops.set_deterministic(True)
    try:
        for fail in (True, False):
            tr = Trainer(sp, device=cuda, compute_dtype=torch.float32, init=False, dropout=True, dropout_seed=5)
            tr.early_catchup = True
            tr.store.load_state(P)
            bs = [tr.make_batch(i, m) for (i, m, _l) in batches]
            for s in range(5):
                if fail and s == 2:
                    real = tr.engine.loss_unbias

                    def boom(*a, **k):
                        raise RuntimeError("injected")
                    tr.engine.loss_unbias = boom
                    with pytest.raises(RuntimeError, match="injected"):
                        tr.train_step(bs[s], prefetch=bs[s + 1])
                    tr.engine.loss_unbias = real
                    assert not tr.opt._begun and tr.opt.global_step == 2
                tr.train_step(bs[s], prefetch=bs[s + 1])
            tr.opt.flush_tables()
            torch.cuda.synchronize()
            res.append((tr.store.tab_p.clone(), tr.store.tab_m.clone(), tr.store.tab_v.clone(), tr.store.params.clone(), tr.opt.global_step))
    finally:
        ops.set_deterministic(False)
    a, b = res
    assert a[4] == b[4] == 5
    for x, y in zip(a[:4], b[:4]):
        assert torch.equal(x, y)


def test_trainer_close_releases_the_last_steps_activations(cuda):
    """An engine's `intermediates` hold graph-attached tensors whose autograd nodes point back at the engine: a cycle through C++ graph edges
    that Python's collector cannot see.  Trainer.close() breaks it -- after close + del the device memory of a Trainer that ran a forward /
    backward pass is back (without close() the pass's activations stay allocated: the AUC tests of tests/test_gpu_configs.py ran the device
    out of memory that way)."""
    import gc
    from cikm2020_dmt_amd import spec as S
    sp = S.scaled_spec(S.e64_spec(), {"Sku": 20000, "Brand": 3000, "Shopid": 3000, "Cid3": 1200})
    inputs, mask, label = make_batch(sp, 1024, seed=3, lengths="full")
    gc.collect(); torch.cuda.synchronize(); torch.cuda.empty_cache()
    base = torch.cuda.memory_allocated(cuda)
    held = []
    for closed in (False, True):
        tr = Trainer(sp, device=cuda, compute_dtype=torch.bfloat16, seed=1, dropout=True)
        b = tr.make_batch(inputs, mask, label)
        float(tr.forward_backward(b))
        (c, o), yb = tr.engine.inference(b)              # (a forward pass whose graph nobody walks: what an evaluation loop leaves behind)
        del c, o, yb, b
        if closed:
            tr.close()
        del tr
        gc.collect(); torch.cuda.synchronize()
        held.append(torch.cuda.memory_allocated(cuda) - base)
        base = torch.cuda.memory_allocated(cuda)
    print("device bytes still allocated after dropping a Trainer: without close() %.1f MB, with close() %.1f MB" % (held[0] / 1e6, held[1] / 1e6))
    assert held[1] < 8e6                                    # (the lane streams' probe buffers and the like: a few MB, once)
    assert held[0] > 20 * max(held[1], 1e6) or held[0] < 8e6      # the leak this guards against (absent only if torch ever learns to collect it)

"""The other five branches of get_optimizer (model/inference_mlp.py:264-280: sgd, adadelta, adagrad, ftrl, rmsprop; adam is
tests/test_gpu_ops.py / test_gpu_boundary.py / test_gpu_model.py), cikm2020_dmt_amd.optim.TFSlotOptimizer over dmt_opt_* :

  * the dense sweep against oracle/dmt_oracle.py:TFOptimizer (TF 1.12's ApplyXxx arithmetic in float64);
  * the sparse-row update with lazily replayed slots against the SAME kernel swept densely over whole tables, as the reference does
    with its densified embedding gradients (run_dnn.py:45-80): var and slots bit for bit for rows idle up to the exact-replay length,
    slots to 1e-5 relative (and the update they feed to 1e-6) beyond; FTRL: rows never read are zero after the first step;
  * three train steps of the whole path per optimizer against the oracle's loss_and_grads + TFOptimizer.
"""
import numpy as np
import pytest
import torch

from oracle import dmt_oracle as O
from oracle import dmt_oracle_torch as OT
from cikm2020_dmt_amd.data_feed.synthetic import make_batch
from cikm2020_dmt_amd.model.inference_mlp import Inference
from cikm2020_dmt_amd.optim import TFAdam, TFSlotOptimizer, make_optimizer
from cikm2020_dmt_amd.train import Trainer
from cikm2020_dmt_amd.variables import VariableStore
from tests.util import small_specs

pytestmark = pytest.mark.gpu
KINDS = list(O.TFOptimizer.KINDS)


@pytest.mark.parametrize("kind", KINDS)
def test_dense_sweep_matches_oracle(cuda, kind):
    _so, sp = small_specs()
    st = VariableStore(sp, cuda, torch.float32, seed=3)
    lr = 0.01
    opt = make_optimizer(kind, st, learning_rate=(lr,), step_boundary=())
    assert isinstance(opt, TFSlotOptimizer)
    rng = np.random.default_rng(5)
    P = {"p": st.params.cpu().numpy().astype(np.float64)}
    p0 = P["p"].copy()
    ref = O.TFOptimizer(kind, lr)
    for step in range(6):
        g = (rng.standard_normal(st.P) * 0.05).astype(np.float32)
        g[rng.random(st.P) < 0.2] = 0.0
        if step == 3:
            g[:] = 0.0
        st.grads.copy_(torch.as_tensor(g))
        opt.begin(); opt.apply_dense(0.5); opt.end()
        ref.apply(P, {"p": g.astype(np.float64) * 0.5})
    got = st.params.cpu().numpy().astype(np.float64)
    moved = np.abs(P["p"] - p0)
    err = np.abs(got - P["p"])
    print(kind, "moved max %.3g, err max %.3g" % (moved.max(), err.max()))
    assert moved.max() > (1e-7 if kind == "adadelta" else 1e-3)
    assert (err <= 2e-5 * moved + 3e-7 * np.maximum(np.abs(P["p"]), 1.0)).all(), float((err / (moved + 1e-12)).max())
    for slot, want in ((st.adam_m, ref.s0["p"]), (st.adam_v, ref.s1["p"])):
        if kind == "sgd" or (kind == "adagrad" and slot is st.adam_v):
            continue
        s = slot.cpu().numpy().astype(np.float64)
        assert np.abs(s - want).max() <= 2e-5 * max(np.abs(want).max(), 1e-12), (kind, float(np.abs(s - want).max()))


def _dense_grads(store, rows, g, cuda):
    out = {}
    for name, (base, nr) in store.table_rows.items():
        dim = store.tables[name].shape[1]
        gd = np.zeros((nr, dim), np.float32)
        sel = (rows >= base) & (rows < base + nr)
        gd[rows[sel] - base] = g[sel][:, :dim]
        out[name] = torch.tensor(gd, device=cuda)
    return out


@pytest.mark.parametrize("kind,wire", [(k, "f32") for k in KINDS] + [("rmsprop", "bf16"), ("ftrl", "bf16")])
def test_lazy_rows_equal_the_dense_table_sweep(cuda, kind, wire):
    _so, sp = small_specs()
    a = VariableStore(sp, cuda, torch.float32, seed=2)
    b = VariableStore(sp, cuda, torch.float32, seed=2)
    oa = make_optimizer(kind, a, learning_rate=(0.01, 0.001), step_boundary=(40,))
    ob = make_optimizer(kind, b, learning_rate=(0.01, 0.001), step_boundary=(40,))
    rng = np.random.default_rng(4)
    D = max(t.shape[1] for t in a.table.values())
    D += (-D) % 4
    pool = np.unique(rng.integers(0, a.total_rows, size=400)).astype(np.int32)     # the rows any batch reads
    rare = pool[:8]                                                                # read at the first and the last step only
    never = np.setdiff1d(np.arange(a.total_rows, dtype=np.int32), pool)
    p_init = a.tab_p.cpu().numpy().copy()
    n_steps = 150
    for step in range(n_steps):
        if step in (0, n_steps - 1):
            rows = pool
        else:
            rows = np.unique(rng.choice(pool[8:], size=60)).astype(np.int32)
        g = (rng.standard_normal((len(rows), D)) * 0.05).astype(np.float32)
        gt = torch.tensor(g, device=cuda)
        if wire == "bf16":
            gt = gt.to(torch.bfloat16)
            g = gt.float().cpu().numpy()
        oa.begin()
        oa.apply_sparse((torch.tensor(rows, device=cuda), torch.tensor([len(rows)], dtype=torch.int32, device=cuda), gt, len(rows)), 0.5)
        oa.end()
        ob.begin(); ob.apply_dense_tables(_dense_grads(b, rows, g, cuda), 0.5); ob.end()
    oa.flush_tables()
    torch.cuda.synchronize()
    pa, pb = a.tab_p.cpu().numpy(), b.tab_p.cpu().numpy()
    # the eight rare rows sat idle for 148 steps: beyond the exact replay length their rmsprop / adadelta slots come from the closed form
    # (1e-5 relative), and the last step's update inherits that; every other element is bit for bit the dense sweep's
    names, bases, dims, offs = a.table_map()
    rare_el = np.zeros(pa.shape, bool)
    for name, base, dim, off in zip(names, bases, dims, offs):
        for r in rare[(rare >= base) & (rare < base + a.table_rows[name][1])] - base:
            rare_el[off + r * dim: off + (r + 1) * dim] = True
    same = pa.view(np.uint32) == pb.view(np.uint32)
    assert same[~rare_el].all(), (kind, float(np.abs(pa - pb)[~rare_el].max()))
    if kind in ("rmsprop", "adadelta"):
        assert np.abs(pa - pb)[rare_el].max() <= 1e-6, (kind, float(np.abs(pa - pb)[rare_el].max()))
    else:
        assert same.all()
    assert not np.array_equal(pa, p_init)
    # rows no batch read: untouched -- except under FTRL, whose first dense step sets them to zero
    for name, base, dim, off in zip(names, bases, dims, offs):
        nr = a.table_rows[name][1]
        sel = never[(never >= base) & (never < base + nr)] - base
        blk = pa[off: off + nr * dim].reshape(nr, dim)[sel]
        want = np.zeros_like(blk) if kind == "ftrl" else p_init[off: off + nr * dim].reshape(nr, dim)[sel]
        assert np.array_equal(blk, want), (kind, name)
    for x, y in ((a.tab_m, b.tab_m), (a.tab_v, b.tab_v)):
        x, y = x.cpu().numpy().astype(np.float64), y.cpu().numpy().astype(np.float64)
        assert np.abs(x - y).max() <= 1e-5 * max(np.abs(y).max(), 1e-30), (kind, float(np.abs(x - y).max()))
    if kind in ("rmsprop", "adadelta"):
        # gaps <= the exact replay length are bit-identical: every pool row but the eight rare ones was read every ~5 steps
        name, base, dim, off = names[0], bases[0], dims[0], offs[0]
        often = pool[8:]
        often = often[(often >= base) & (often < base + a.table_rows[name][1])] - base
        if len(often):
            xa = a.tab_m.cpu().numpy()[off:].reshape(-1)[: a.table_rows[name][1] * dim].reshape(-1, dim)[often]
            xb = b.tab_m.cpu().numpy()[off:].reshape(-1)[: a.table_rows[name][1] * dim].reshape(-1, dim)[often]
            assert np.array_equal(xa.view(np.uint32), xb.view(np.uint32))


@pytest.mark.parametrize("kind", KINDS)
def test_train_steps_with_each_optimizer_match_oracle_fp32(cuda, kind):
    so, sp = small_specs()
    P = O.init_params(so, seed=11)
    lr = 0.01
    tr = Trainer(sp, device=cuda, compute_dtype=torch.float32, init=False, dropout=False, optimizer=kind, learning_rate=(lr,), step_boundary=())
    tr.store.load_state(P)
    Pn = {k: v.copy() for k, v in P.items()}
    ref = O.TFOptimizer(kind, lr)
    batches = [make_batch(sp, 16, seed=100 + i, lengths="ragged", weights="random") for i in range(3)]
    for (inp, m, _l) in batches:
        lref, _lg, G = OT.loss_and_grads(Pn, inp, m, so)
        ref.apply(Pn, G)
        lgot = float(tr.train_step(tr.make_batch(inp, m)))
        assert abs(lgot - lref) <= 2e-4 * abs(lref), (kind, lgot, lref)
    tr.opt.flush_tables()
    got = tr.store.state_dict()
    worst = []
    for k in Pn:
        moved = float(np.abs(Pn[k] - P[k]).max())
        err = float(np.abs(got[k] - Pn[k]).max())
        worst.append((err, moved, k))
    # (the K bias of an attention block has no gradient in exact arithmetic -- it shifts every score of a softmax alike --: what moves it
    #  is rounding noise, 1e-19 here; errors are measured against the tensor's own move or 1e-4 of the largest move of any tensor)
    top = max(w[1] for w in worst)
    worst = [(e / max(mv, 1e-4 * top), e, mv, k) for (e, mv, k) in worst]
    worst.sort(reverse=True)
    print(kind, "worst (error / largest move of the tensor, error, move):", worst[:4])
    # gradients agree with the oracle's to ~2e-3 of the tensor's norm (tests/test_gpu_model.py), and every update rule here is smooth
    # in the gradient at the sizes that occur -- adadelta's first steps excepted: update = g sqrt(eps) / sqrt((1-rho) g^2 + eps) is
    # sign(g) * 4.5e-4 for |g| >> 4e-4 and lr = 0.01 of that, so an element whose gradient is rounding-sized may go either way
    bound = 0.02 if kind != "adadelta" else 2.0
    assert worst[0][0] <= bound, worst[:5]
    if kind == "ftrl":
        # the rows no batch has read are zero (dense ApplyFtrl with linear == 0), in the oracle and here
        name = max(tr.store.tables, key=lambda n: tr.store.tables[n].shape[0])
        zero_ref = np.abs(Pn[name]).max(axis=1) == 0
        assert zero_ref.any() and (np.abs(got[name][zero_ref]).max() == 0)


def test_flush_before_the_first_step_changes_nothing(cuda):
    """A checkpoint written before any step flushes the lazy rows first: with fresh slots that must be a no-op for every kind (an FTRL
    sweep over linear == 0 would zero the tables)."""
    _so, sp = small_specs()
    for kind in KINDS:
        st = VariableStore(sp, cuda, torch.float32, seed=4)
        before = st.tab_p.clone()
        opt = make_optimizer(kind, st)
        opt.flush_tables()
        opt.reset_slots(1000)
        opt.flush_tables()
        torch.cuda.synchronize()
        assert torch.equal(st.tab_p, before), kind


def test_get_optimizer_returns_every_reference_branch(cuda):
    _so, sp = small_specs()
    inf = Inference(None, device=cuda, compute_dtype=torch.float32, seed=4, spec=sp)
    for name in ("sgd", "adadelta", "adagrad", "ftrl", "rmsprop"):
        o = inf.get_optimizer(name, 0.001)
        assert isinstance(o, TFSlotOptimizer) and o.kind == name
    assert isinstance(inf.get_optimizer("adam", 0.001), TFAdam)
    with pytest.raises(SystemExit):
        inf.get_optimizer("lamb", 0.001)

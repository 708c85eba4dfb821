"""The BENCHMARKED configuration (BASELINE.json configs[1]: E64 = every id field 64 wide, d_model 320, d_ff 1280, 4 heads of 80, bf16)
against the CPU oracle END TO END, with every default switch on -- so the kernels that are only dispatched at these dims run inside a
model-level oracle comparison: dmt_mhsa_block_fwd (the fused self-attention block; with it off: dmt_proj, the 320 -> 960 streamed-weight
QKV projection, and attn_fwd_co_kernel<80>), attn_bwd_co_kernel<80>, chain2<320,1280,320>,
dmt_wgrad320, dmt_q1mem_fwd/bwd, the fused MMoE expert kernels and the fused heads.  Each test records the launch routes the library
took (dmt_route_trace) and asserts them: a silent fall back to the generic kernels would fail the test, not pass it.

Reference: TransformerModel_util.py:160-209 (multihead_attention), :212-235 (ff), mmoe_transformer_unbias.py:130-186 (generate_data),
:63-126 (expert_gate, build_tower), base.py:93-134 (embedding_combiner), inference_mlp.py:162-223 (loss).
Vocabularies are scaled down (the oracle holds dense fp64 tables); the model dims are the benchmarked ones.

Tolerances (bf16 activations, fp32 accumulation, fp64 oracle): logits 6e-2 absolute (values are O(1)), loss 3e-2 relative, gradient
L2 error <= 0.2 of a tensor's L2 norm and element-wise ulp bands as in tests/test_gpu_model.py; fp32 mode: 3e-4 / 1e-5 / 3e-3.
"""
import numpy as np
import pytest
import torch

from oracle import dmt_oracle as O
from oracle import dmt_oracle_torch as OT
from cikm2020_dmt_amd import _lib as L
from cikm2020_dmt_amd import ops
from cikm2020_dmt_amd import spec as S
from cikm2020_dmt_amd.data_feed.synthetic import make_batch
from cikm2020_dmt_amd.spec import trans_prefix
from cikm2020_dmt_amd.train import Trainer
from tests.util import sparse_to_dense_tables

pytestmark = pytest.mark.gpu

E64_ROWS = {"Sku": 2000, "Brand": 300, "Shopid": 300, "Cid3": 120, "Cid2": 50}
# routes the E64 bf16 step must take (labels of DMT_CHECK_LAUNCH in csrc/)
# (fused self-attention block on, the default; E64_ROUTES_UNFUSED: Trainer(fused_mhsa=False), the three-launch self-attention forward)
E64_ROUTES = ("dmt_gather_fwd", "dmt_mhsa_block_fwd", "dmt_attn_bwd(mfma, coalesced)", "dmt_chain2", "dmt_wgrad320",
              "dmt_q1mem_fwd", "dmt_q1mem_bwd", "dmt_mmoe_experts_fwd(split)", "dmt_mmoe_experts_bwd(split)", "dmt_heads_fwd", "dmt_heads_bwd",
              "dmt_embgrad_reduce")
E64_ROUTES_UNFUSED = tuple(r for r in E64_ROUTES if r != "dmt_mhsa_block_fwd") + ("dmt_proj", "dmt_attn_fwd(mfma, coalesced)")
BF16_MAX_ULPS, BF16_TAIL_ULPS, BF16_TAIL_FRAC = 256.0, 16.0, 0.12
BF16_FAR_ULPS, BF16_FAR_FRAC = 64.0, 0.003
TOL = {torch.float32: dict(logit=3e-4, loss=1e-5, grad=3e-3, floor=1e-6), torch.bfloat16: dict(logit=6e-2, loss=3e-2, grad=0.2, floor=3e-3)}


def e64_specs():
    sp = S.scaled_spec(S.e64_spec(), E64_ROWS)
    assert sp["d_model"] == 320 and sp["d_ff"] == 1280 and sp["num_heads"] == 4
    return dict(sp), sp          # the oracle reads the same dict (spec-generic)


def _params(so, seed=11):
    P = O.init_params(so, seed=seed)
    rng = np.random.default_rng(3)
    for k in P:          # non-trivial LayerNorm / bias parameters, so their use and their gradients are exercised
        if k.endswith("/gamma"):
            P[k] = P[k] + 0.1 * rng.standard_normal(P[k].shape)
        if k.endswith("/beta") or k.endswith("/bias") or k.endswith("biases"):
            P[k] = P[k] + 0.05 * rng.standard_normal(P[k].shape)
    return P


def _setup(cuda, dtype, B, seed=5, lengths="ragged", weights="random", dropout=False, dropout_seed=1, fused_mhsa=None, packed_rows=False):
    """packed_rows False: the dense [B, T, d] layout (what these tests covered before round 5); True: sequences with >= 10 % padding run
    on their real rows only (engine.SeqPack) -- the same oracle comparison then covers the packed forms of the kernels."""
    so, sp = e64_specs()
    P = _params(so)
    inputs, mask, label = make_batch(sp, B, seed=seed, lengths=lengths, weights=weights)
    tr = Trainer(sp, device=cuda, compute_dtype=dtype, init=False, dropout=dropout, dropout_seed=dropout_seed, fused_mhsa=fused_mhsa,
                 packed_rows=packed_rows)
    tr.store.load_state(P)
    return so, sp, P, inputs, mask, tr, tr.make_batch(inputs, mask, label)


def _check_grads(tr, G, dtype, label=""):
    t = TOL[dtype]
    got = dict(tr.store.grad_dict())
    got.update(sparse_to_dense_tables(tr.store, tr.engine.sparse))
    assert set(got) == set(G)
    gscale = max(np.abs(G[n]).max() for n in got)
    bad, worst_l2, worst_ulps = [], [], {}
    for name, g in got.items():
        ref = G[name]
        denom = max(np.linalg.norm(ref), t["floor"] * gscale * np.sqrt(ref.size))
        e = float(np.linalg.norm(g - ref) / denom)
        worst_l2.append((e, name))
        if not e < t["grad"]:
            bad.append((name, "L2", e))
        if dtype == torch.bfloat16:
            tscale = max(float(np.abs(ref).max()), 1e-2 * gscale)
            ulps = np.abs(g - ref).reshape(-1) / (tscale * 2.0 ** -8)
            frac, far = float((ulps > BF16_TAIL_ULPS).mean()), float((ulps > BF16_FAR_ULPS).mean())
            worst_ulps[name] = (float(ulps.max()), frac)
            if not (ulps.max() < BF16_MAX_ULPS and frac <= BF16_TAIL_FRAC + 1.0 / ulps.size and far <= BF16_FAR_FRAC + 1.0 / ulps.size):
                bad.append((name, "bf16 element errors (max ulps, beyond tail, beyond far)", float(ulps.max()), frac, far))
        else:
            worst = float(np.abs(g - ref).max())
            if not worst < 1e-3 * max(float(np.abs(ref).max()), 1e-3 * gscale):
                bad.append((name, "max element error", worst, float(np.abs(ref).max())))
    print("%s worst L2 gradient errors:" % label, sorted(worst_l2, reverse=True)[:4])
    if worst_ulps:
        print("%s worst element errors (ulps of the tensor scale, fraction beyond %g):" % (label, BF16_TAIL_ULPS),
              sorted(worst_ulps.items(), key=lambda kv: -kv[1][0])[:4])
    assert not bad, "gradient mismatches: %s" % bad


def _assert_routes(counts, wanted=E64_ROUTES):
    # (a ragged batch runs on packed rows by default -- engine.SeqPack --: the fused block then reports its packed form)
    alias = {"dmt_mhsa_block_fwd": ("dmt_mhsa_block_fwd(packed)",)}
    missing = [r for r in wanted if counts.get(r, 0) == 0 and not any(counts.get(a, 0) for a in alias.get(r, ()))]
    assert not missing, "kernels this test claims to cover were not dispatched: %s (took: %s)" % (missing, sorted(counts))


@pytest.mark.parametrize("B,lengths,weights,fused", [(24, "ragged", "random", True), (24, "full", "ones", True), (352, "ragged", "random", True),
                                                     (24, "ragged", "random", False), (352, "ragged", "random", False),
                                                     (24, "ragged", "random", "packed"), (352, "ragged", "random", "packed")])
def test_e64_bf16_forward_loss_and_every_gradient_match_oracle(cuda, monkeypatch, B, lengths, weights, fused):
    """B = 352: the L = 50 sequences have M = 17600 >= WGRAD320_MIN_ROWS rows, so every dispatch rule is the benchmark's own;
    B = 24: the row threshold of the wide-block weight-gradient kernel is lowered so it still runs."""
    packed = fused == "packed"       # "packed": the fused block on packed rows (round 5), against the same oracle
    if B * 50 < ops.WGRAD320_MIN_ROWS or packed:
        # (packed rows: about half of B * 50 rows exist -- the wide-block weight-gradient kernel must still be the one that runs)
        monkeypatch.setattr(ops, "WGRAD320_MIN_ROWS", 256 if (packed and B * 50 < ops.WGRAD320_MIN_ROWS) else 1024)
    dtype = torch.bfloat16
    fused = bool(fused)
    so, sp, P, inputs, mask, tr, batch = _setup(cuda, dtype, B, lengths=lengths, weights=weights, fused_mhsa=fused, packed_rows=packed)
    loss_ref, (c_ref, o_ref, yb_ref), G = OT.loss_and_grads(P, inputs, mask, so)
    if B <= 24:        # the literal numpy restatement agrees with the torch one (the two oracles, at these dims)
        (c2, o2), yb2 = O.inference(inputs, P, so)
        assert np.abs(c2 - c_ref).max() < 1e-9 and np.abs(o2 - o_ref).max() < 1e-9 and np.abs(yb2 - yb_ref).max() < 1e-9
    with L.route_trace() as rt:
        loss = tr.forward_backward(batch)
        torch.cuda.synchronize()
    if packed:
        assert all(tr.engine.intermediates["pack_%d" % i] is not None for i in range(3))
        _assert_routes(rt.counts, tuple(r for r in E64_ROUTES if r != "dmt_mhsa_block_fwd") + ("dmt_mhsa_block_fwd(packed)", "dmt_colsum_rows_packed"))
        assert rt.counts.get("dmt_mhsa_block_fwd", 0) == 0
    else:
        _assert_routes(rt.counts, E64_ROUTES if fused else E64_ROUTES_UNFUSED)
        assert rt.counts.get("dmt_mhsa_block_fwd(packed)", 0) == 0
    (c, o), yb = tr.last["out"]
    t = TOL[dtype]
    errs = [np.abs(x.detach().float().cpu().numpy() - r).max() for x, r in ((c, c_ref), (o, o_ref), (yb, yb_ref))]
    print("E64 bf16 B=%d max |dlogit| click/order/bias:" % B, errs, "loss rel", abs(float(loss) - loss_ref) / abs(loss_ref))
    assert max(errs) < t["logit"]
    assert abs(float(loss) - loss_ref) / abs(loss_ref) < t["loss"]
    pc_ref, pv_ref = O.cal_ctr_cvr_unbias((c_ref, o_ref), yb_ref)
    assert np.abs(tr.last["p_ctr"].cpu().numpy() - pc_ref.reshape(-1)).max() < t["logit"]
    assert np.abs(tr.last["p_cvr"].cpu().numpy() - pv_ref.reshape(-1)).max() < t["logit"]
    _check_grads(tr, G, dtype, "E64 bf16 B=%d" % B)


@pytest.mark.parametrize("packed", [False, True])
def test_e64_bf16_two_encoder_and_two_decoder_blocks_match_oracle(cuda, monkeypatch, packed):
    """num_blocks_encode = num_blocks_decode = 2 (TransformerModel.py:104-121, 154-169; dmt.conf has 1 + 1) through the E64 kernels: two
    fused self-attention blocks and fused feed-forwards back to back per sequence (the second on the first's output, packed rows or dense),
    two raw-memory decoder blocks; forward, loss and every gradient against the oracle, every fused route taken twice as often."""
    monkeypatch.setattr(ops, "WGRAD320_MIN_ROWS", 256)
    dtype = torch.bfloat16
    so, sp = e64_specs()
    so = sp = dict(sp, num_blocks_encode=2, num_blocks_decode=2)
    P = _params(so)
    inputs, mask, label = make_batch(sp, 24, seed=6, lengths="ragged", weights="random")
    tr = Trainer(sp, device=cuda, compute_dtype=dtype, init=False, dropout=False, packed_rows=packed)
    tr.store.load_state(P)
    loss_ref, (c_ref, o_ref, yb_ref), G = OT.loss_and_grads(P, inputs, mask, so)
    with L.route_trace() as rt:
        loss = tr.forward_backward(tr.make_batch(inputs, mask, label))
        torch.cuda.synchronize()
    key = "dmt_mhsa_block_fwd(packed)" if packed else "dmt_mhsa_block_fwd"
    assert rt.counts.get(key, 0) == 6 and rt.counts.get("dmt_q1mem_fwd", 0) == 6 and rt.counts.get("dmt_chain2", 0) == 24, sorted(rt.counts.items())
    (c, o), yb = tr.last["out"]
    t = TOL[dtype]
    errs = [np.abs(x.detach().float().cpu().numpy() - r).max() for x, r in ((c, c_ref), (o, o_ref), (yb, yb_ref))]
    print("E64 bf16 2+2 blocks (packed %s): max |dlogit|" % packed, errs)
    assert max(errs) < 1.5 * t["logit"] and abs(float(loss) - loss_ref) / abs(loss_ref) < t["loss"]      # (two blocks deep: 1.5x the one-block bound)
    got = dict(tr.store.grad_dict())
    got.update(sparse_to_dense_tables(tr.store, tr.engine.sparse))
    assert set(got) == set(G)
    gscale = max(np.abs(G[n]).max() for n in got)
    worst = sorted(((float(np.linalg.norm(g - G[n]) / max(np.linalg.norm(G[n]), t["floor"] * gscale * np.sqrt(g.size))), n) for n, g in got.items()), reverse=True)
    print("worst gradient distances:", worst[:3])
    assert worst[0][0] < 0.3, worst[:5]


def test_e64_fp32_mode_matches_oracle(cuda):
    """The same dims in fp32 mode (generic GEMM, scalar attention kernels at d_h = 80): tight tolerances."""
    dtype = torch.float32
    so, sp, P, inputs, mask, tr, batch = _setup(cuda, dtype, 20)
    loss_ref, (c_ref, o_ref, yb_ref), G = OT.loss_and_grads(P, inputs, mask, so)
    loss = tr.forward_backward(batch)
    (c, o), yb = tr.last["out"]
    t = TOL[dtype]
    assert np.abs(c.detach().cpu().numpy() - c_ref).max() < t["logit"]
    assert np.abs(o.detach().cpu().numpy() - o_ref).max() < t["logit"]
    assert np.abs(yb.detach().cpu().numpy() - yb_ref).max() < t["logit"]
    assert abs(float(loss) - loss_ref) / abs(loss_ref) < t["loss"]
    _check_grads(tr, G, dtype, "E64 fp32")


def test_e64_bf16_with_dropout_matches_oracle(cuda, monkeypatch):
    """Train-mode dropout (0.1 block input + attention weights, 0.5 bias tower) through the E64 kernels: the counter mask is restated by
    the oracle, so the comparison runs with dropout ON (the benchmark's mode)."""
    monkeypatch.setattr(ops, "WGRAD320_MIN_ROWS", 1024)
    dtype = torch.bfloat16
    so, sp, P, inputs, mask, tr, batch = _setup(cuda, dtype, 24, seed=4, dropout=True, dropout_seed=123)
    so = dict(so, dropout_rate=0.1, dropout_rate_bias=[0.5, 0.5])
    loss_ref, (c_ref, o_ref, yb_ref), G = OT.loss_and_grads(P, inputs, mask, so, step_seed=123)
    _l0, (c0, _o0, _y0), _G0 = OT.loss_and_grads(P, inputs, mask, dict(so, dropout_rate=0.0, dropout_rate_bias=[0.0, 0.0]))
    assert np.abs(c_ref - c0).max() > 1e-3
    with L.route_trace() as rt:
        loss = tr.forward_backward(batch)
        torch.cuda.synchronize()
    _assert_routes(rt.counts)
    (c, o), yb = tr.last["out"]
    errs = [np.abs(x.detach().float().cpu().numpy() - r).max() for x, r in ((c, c_ref), (o, o_ref), (yb, yb_ref))]
    print("E64 bf16 dropout max |dlogit|:", errs)
    assert max(errs) < 8e-2
    assert abs(float(loss) - loss_ref) / abs(loss_ref) < 3e-2
    _check_grads(tr, G, dtype, "E64 bf16 dropout")


def test_e64_bf16_three_train_steps_match_oracle(cuda, monkeypatch):
    """3 optimizer steps (dense Adam + exact lazy rows) in bf16 mode against the oracle's dense TFAdam in float64.  Adam normalises the
    gradient, so an element moves by ~lr per step whatever its gradient's size: elements whose true gradient is below the bf16 noise
    floor may move the other way (<= 2 * 3 * lr apart); what must hold: the losses agree step by step, the BULK of every tensor
    agrees to a small fraction of the distance moved, and nothing is further apart than 2 * 3 * lr."""
    monkeypatch.setattr(ops, "WGRAD320_MIN_ROWS", 1024)
    so, sp, P, inputs, mask, tr, batch = _setup(cuda, torch.bfloat16, 24)
    Pn = {k: v.copy() for k, v in P.items()}
    adam = O.TFAdam(lr=1e-3)
    batches = [make_batch(sp, 24, seed=200 + i, lengths="ragged", weights="random") for i in range(3)]
    with L.route_trace() as rt:
        for (inp, m, _l) in batches:
            lref, _lg, G = OT.loss_and_grads(Pn, inp, m, so)
            adam.apply(Pn, G)
            lgot = float(tr.train_step(tr.make_batch(inp, m)))
            assert abs(lgot - lref) / abs(lref) < 3e-2, (lgot, lref)
        tr.opt.flush_tables()
        torch.cuda.synchronize()
    _assert_routes(rt.counts)
    got = tr.store.state_dict()
    lr3 = 3e-3
    stats = []
    for k in Pn:
        moved = np.abs(Pn[k] - P[k])
        sel = moved > 0.5 * lr3                    # elements the oracle really moved (rows never touched stay put in both)
        dev = np.abs(got[k] - Pn[k])
        assert dev.max() <= 2 * lr3 * 1.05, (k, float(dev.max()))
        if sel.sum() >= 16:
            stats.append((float(np.median(dev[sel])), float((dev[sel] > 0.25 * lr3).mean()), k))
        if k in tr.store.tables:
            untouched = moved.max(axis=1) == 0
            if untouched.any():
                assert float(dev[untouched].max()) < 1e-6, k      # lazy Adam: rows no batch read equal their initial values
    print("E64 bf16 3 steps: worst median deviation:", sorted(stats, reverse=True)[:3], "| worst fraction beyond lr*0.75:",
          sorted(stats, key=lambda s: -s[1])[:3])
    assert max(s[0] for s in stats) < 0.05 * lr3, sorted(stats, reverse=True)[:3]
    assert max(s[1] for s in stats) < 0.10, sorted(stats, key=lambda s: -s[1])[:3]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_e64_gather_and_pool_values_match_oracle(cuda, dtype):
    """generate_data (mmoe_transformer_unbias.py:130-186) and embedding_combiner (base.py:93-134) at E64: the gather's sequence rows,
    target rows (through the input prep: * sqrt(d) + learned positions, TransformerModel.py:96-100) and pooled columns, value by value."""
    so, sp, P, inputs, mask, tr, batch = _setup(cuda, dtype, 33)
    eng = tr.engine
    d = sp["d_model"]
    tol = 1e-6 if dtype == torch.float32 else 2.0 ** -8
    ref = O.generate_data(inputs, P, so)
    # raw lookups ([0;E] semantics, no scale, no positions)
    Xr, tar_r = eng.gather_raw(batch)
    for s, (mk_r, ln_r, se_r, ta_r, _ts) in enumerate(ref):
        T = se_r.shape[1]
        got = Xr[s].float().cpu().numpy()[:, :T]
        assert np.abs(got - se_r).max() <= tol * np.abs(se_r).max() + 1e-7, s
        assert np.abs(tar_r.float().cpu().numpy() - ta_r).max() <= tol * np.abs(ta_r).max() + 1e-7
    # the training gather: scaled + positions, zeros past the length; pooled | dense columns of z
    with torch.no_grad():
        X, tar, z = eng.gather(batch)
    for s, (mk_r, ln_r, se_r, ta_r, _ts) in enumerate(ref):
        T = se_r.shape[1]
        pos = P[trans_prefix(s) + "positional_encoding_k_position_learn/embedding_position_learn"][:T]
        want = se_r * np.sqrt(d) + pos[None]
        got = X[s].float().cpu().numpy()[:, :T]
        valid = mk_r.astype(bool)
        assert np.abs(got - want)[valid].max() <= 2 * tol * np.abs(want).max() + 1e-6, s
        assert np.abs(tar.float().cpu().numpy() - ta_r * np.sqrt(d)).max() <= 2 * tol * np.abs(ta_r * np.sqrt(d)).max() + 1e-6
    comb = O.embedding_combiner(inputs, P, so)                         # [B, 615 + sum dims]
    zc = z.float().cpu().numpy()[:, : comb.shape[1]]
    assert np.abs(zc - comb).max() <= 2 * tol * np.abs(comb).max() + 1e-6
    bias_ref = O.embedding_combiner(inputs, P, so, emb_list=so["embedding_list_bias"], prefix="", with_dense=False)
    zb = z.float().cpu().numpy()[:, eng.plan.bias_off: eng.plan.bias_off + eng.plan.bias_width]
    assert np.abs(zb - bias_ref).max() <= 2 * tol * np.abs(bias_ref).max() + 1e-6


@pytest.mark.parametrize("drop", [False, True])
@pytest.mark.parametrize("B,T", [(7, 10), (33, 50), (5, 64), (300, 50)])
def test_e64_raw_memory_decoder_attention_matches_oracle_directly(cuda, B, T, drop):
    """dmt_q1mem (the decoder's cross attention over the RAW memory rows: K / V projections re-associated away, bias folded into
    column 320, a softmax-invariant constant dropped) against the oracle's multihead_attention (TransformerModel_util.py:160-209)
    itself -- forward and every gradient (query input, memory, packed kernel / bias, LayerNorm) -- not against a sibling kernel."""
    so, sp = e64_specs()
    P = _params(so, seed=4)
    rng = np.random.default_rng(B * 100 + T)
    blk = trans_prefix(1) + "num_blocks_0/"
    a = blk + "vanilla_attention/"
    for nm in ("dense", "dense_1", "dense_2"):
        P[a + nm + "/bias"] = 0.2 * rng.standard_normal(P[a + nm + "/bias"].shape)
    tr = Trainer(sp, device=cuda, compute_dtype=torch.bfloat16, init=False, dropout=False)
    tr.store.load_state(P)
    eng, st = tr.engine, tr.store
    d, H = sp["d_model"], sp["num_heads"]
    bf = lambda x: torch.tensor(x).to(torch.bfloat16)
    y0, mem0 = bf(rng.standard_normal((B, 1, d)) * 0.8), bf(rng.standard_normal((B, T, d)) * 0.8)
    lens_np = rng.integers(1, T + 1, size=B)
    lens_np[0] = T
    if B > 2:
        lens_np[1] = 1
    w = rng.standard_normal((B, 1, d))
    seed = 77 if drop else None
    # oracle (fp64, on the bf16-rounded inputs)
    Pt = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in P.items() if k.startswith(a)}
    yq = y0.double().requires_grad_(True)
    mq = mem0.double().requires_grad_(True)
    ref = OT._mha(yq, mq, torch.ones(B, dtype=torch.long), torch.tensor(lens_np), H, Pt, a, rate=0.1 if drop else 0.0, step_seed=seed, stream=13)
    (ref * torch.tensor(w)).sum().backward()
    # HIP
    eng.dropout_step_seed = seed
    assert eng.use_q1mem and a in st.q1mem
    st.zero_grad()
    y = y0.to(cuda).requires_grad_(True)
    mem = mem0.to(cuda).requires_grad_(True)
    with L.route_trace() as rt:
        s = eng.mha_cross(y, mem, None, torch.tensor(lens_np, dtype=torch.int32, device=cuda), blk, 13)
        (s.float() * torch.tensor(w, dtype=torch.float32, device=cuda)).sum().backward()
        torch.cuda.synchronize()
    _assert_routes(rt.counts, ("dmt_q1mem_fwd", "dmt_q1mem_bwd"))

    def close(got, want, rel, what):
        got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
        e = np.abs(got - want).max() / (np.abs(want).max() + 1e-30)
        l2 = np.linalg.norm(got - want) / (np.linalg.norm(want) + 1e-30)
        assert e < rel and l2 < rel / 2, (what, e, l2)

    close(s.detach().float().cpu().numpy(), ref.detach().numpy(), 3e-2, "output")
    close(y.grad.float().cpu().numpy(), yq.grad.numpy(), 3e-2, "d query input")
    dm, dm_ref = mem.grad.float().cpu().numpy(), mq.grad.numpy()
    close(dm, dm_ref, 3e-2, "d memory")
    kmask = np.arange(T)[None, :] >= lens_np[:, None]
    assert float(np.abs(dm[kmask]).max() if kmask.any() else 0.0) == 0.0            # masked keys get no gradient
    G = st.grad_dict()
    # the softmax is invariant to the key bias: its true gradient is 0 -- bounded against the query bias's scale instead
    gq = Pt[a + "dense/bias"].grad.numpy()
    for nm in ("dense/kernel", "dense/bias", "dense_1/kernel", "dense_2/kernel", "dense_2/bias", "ln/gamma", "ln/beta"):
        close(G[a + nm], Pt[a + nm].grad.numpy(), 4e-2, nm)
    assert np.abs(G[a + "dense_1/bias"]).max() < 2e-2 * np.abs(gq).max()

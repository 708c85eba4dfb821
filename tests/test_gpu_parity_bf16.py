"""The benchmarked bf16 mode against the STORAGE-ROUNDING oracle (oracle/dmt_oracle_torch.py, storage="bf16"): the same float64
restatement of the reference, with every tensor the HIP engine keeps in bf16 rounded to bf16 where the engine stores it (activations
and their gradients, the bf16 weight copies; sums still in float64).  What is left between the two is accumulation order and fp32
accumulators, so the gradient bound is an order of magnitude below the one against the unrounded oracle (tests/test_gpu_e64.py: 20 %
of a tensor's norm, measured 7-12 %): a defect worth a few per cent of a gradient tensor -- one dropped term of a weight gradient, a
wrong LayerNorm statistic -- fails here.  The unrounded comparison stays the accuracy statement.

The bound is CALIBRATED per tensor, not granted.  Legitimate evaluations of the very same rounded function -- this oracle with
float64 sums, with float32 sums, with its parameters moved by half an fp32 ulp -- already differ: a
perturbation far below a bf16 ulp flips a few roundings, every flip perturbs what follows by a full bf16 ulp, and after three or
four storage points the difference has grown to the bf16 noise level whatever its origin (scripts/parity_bf16_stages.py: the
encoder memories of two such evaluations differ by 5e-4, the logits by 3e-3 -- and the HIP engine differs from either by EXACTLY
those amounts at every stage; its bias-tower logit agrees to 3e-8).  In the gradients that is, on MI355X's host, B = 24: median
1.1 %, 90th percentile 2.0 %, worst tensor 6.8 % (`mmoe_layers/expert-2/expert-layer-2/weights`); B = 352: 0.8 % / 1.3 % / 4.0 %;
the HIP engine (fp32 accumulators, its own summation order: one more member of the family) lands at 1.1 % / 2.0 % / 6.8 % with the
same worst tensor.  That is the floor ANY correct bf16 implementation has against a given oracle: a flat "<= 2 % for every tensor"
cannot be met by any of them, and a flat 7 % would hide a 5 % defect in a tensor whose floor is 1 %.  Asserted instead, per tensor,
with floor = the largest distance of an ensemble member (float32 sums; float32 sums of half-ulp-perturbed parameters) from the float64 oracle:
    err(HIP, oracle64)  <=  2 * floor + 0.01      (<= 0.10 in any case; tensors of fewer than 8 elements: at least 0.04 -- a scalar
                                                    gradient such as an output bias is ONE cancelling sum, its floor one random draw)
and the median over the tensors <= 1.5 %.  Each test prints how many of the 121 tensors a systematic 3 % / 5 % error would fail
(the check's power); scripts/parity_loss_heads.py shows the loss kernel exact to 5e-8 and the towers' bias gradients bit-equal to
the sum of the bf16-rounded logit gradients, which is what the oracle's storage model states.

  * B = 24 (ragged / full, dropout on and off) and B = 352 (every dispatch rule is the benchmark's own), E64 dims, every default
    kernel, launch routes asserted;
  * BASELINE configs[1] AT ITS REAL SIZE: B = 4096, L = 50/50/10, E64, bf16 -- one forward + backward against the oracle (both
    forms): logits, loss, every dense gradient, every embedding-gradient row (small vocabularies, so the oracle's dense tables
    are light); the grid-size-dependent paths (XCD remap over 1 600 tiles, wgrad320 row splits over 204 800 rows, 31 output tiles
    per persistent workgroup) run inside an oracle comparison here.

Reference: TransformerModel_util.py:160-235, mmoe_transformer_unbias.py:63-233, base.py:93-134, inference_mlp.py:162-223.
Parity is UNPINNED (no TensorFlow, no reference goldens: oracle/dmt_oracle.py header): green here means "partial".
"""
import os

import numpy as np
import pytest
import torch

from oracle import dmt_oracle as O
from oracle import dmt_oracle_torch as OT
from cikm2020_dmt_amd import _lib as L
from cikm2020_dmt_amd import ops
from cikm2020_dmt_amd import spec as S
from cikm2020_dmt_amd.data_feed.synthetic import make_batch
from cikm2020_dmt_amd.train import Trainer
from tests.test_gpu_e64 import E64_ROWS, _assert_routes, _params
from tests.util import sparse_to_dense_tables

pytestmark = pytest.mark.gpu

# measured on MI355X (scripts/parity_bf16_report.py; the numbers are printed by every run of these tests):
#   gradients, L2 error / tensor norm, worst tensor:   B = 24: see TOL_SMALL   B = 352 / 4096: see TOL
#   logits: max |d| 0.009-0.022 (values are O(1); one bf16 ulp at 2 is 0.008)
TOL = dict(logit=3e-2, loss=2e-3, grad_median=0.015, cap=0.10)
FLOOR = 3e-3       # tensors whose true gradient is (numerically) zero -- the key bias of a softmax -- are measured against the largest gradient


def _dist(got, G):
    gscale = max(np.abs(np.asarray(G[n])).max() for n in got)
    out = {}
    for name, g in got.items():
        ref = np.asarray(G[name], dtype=np.float64)
        denom = max(np.linalg.norm(ref), FLOOR * gscale * np.sqrt(ref.size))
        out[name] = float(np.linalg.norm(np.asarray(g, dtype=np.float64) - ref) / denom)
    return out


def _hip_grads(tr):
    got = dict(tr.store.grad_dict())
    got.update(sparse_to_dense_tables(tr.store, tr.engine.sparse))
    return got


def _run(cuda, B, lengths, weights, dropout, seed=5):
    sp = S.scaled_spec(S.e64_spec(), E64_ROWS)
    so = dict(sp)
    P = _params(so)
    inputs, mask, label = make_batch(sp, B, seed=seed, lengths=lengths, weights=weights)
    step_seed = 123 if dropout else None
    tr = Trainer(sp, device=cuda, compute_dtype=torch.bfloat16, init=False, dropout=dropout, dropout_seed=step_seed or 1)
    tr.store.load_state(P)
    batch = tr.make_batch(inputs, mask, label)
    if dropout:
        so = dict(so, dropout_rate=0.1, dropout_rate_bias=[0.5, 0.5])
    with L.route_trace() as rt:
        loss = float(tr.forward_backward(batch))
        torch.cuda.synchronize()
    _assert_routes(rt.counts)
    return tr, loss, (P, inputs, mask, so, step_seed)


def _ensemble(P, inputs, mask, so, step_seed, n_perturbed=1):
    """Other legitimate evaluations of the storage-rounded function, of the HIP engine's own arithmetic class: float32 sums, and
    float32 sums with every parameter moved by half an fp32 ulp (3e-8 relative: what rounding the inputs of a sum once more does).
    (Measured: a 3e-7 perturbation -- five fp32 ulps -- already triples the distances; the family is the fp32-rounding-sized one.)"""
    out = [OT.loss_and_grads(P, inputs, mask, so, step_seed=step_seed, storage="bf16", dtype=torch.float32)]
    for k in range(n_perturbed):
        rng = np.random.default_rng(1000 + k)
        Pk = {n: v * (1.0 + 3e-8 * rng.standard_normal(v.shape)) for n, v in P.items()}
        out.append(OT.loss_and_grads(Pk, inputs, mask, so, step_seed=step_seed, storage="bf16", dtype=torch.float32))
    return out


def _compare_calibrated(tr, loss, ref64, ensemble, label, tol=TOL, min_power=0.5):
    """ref64: loss_and_grads of the storage-rounding oracle with float64 sums; ensemble: _ensemble(...)."""
    lref, (c_ref, o_ref, yb_ref), G64 = ref64
    (c, o), yb = tr.last["out"]
    dl = max(float(np.abs(x.detach().float().cpu().numpy() - r).max()) for x, r in ((c, c_ref), (o, o_ref), (yb, yb_ref)))
    lrel = abs(loss - lref) / abs(lref)
    got = _hip_grads(tr)
    assert set(got) == set(G64)
    err = _dist(got, G64)
    dists = [_dist(m[2], G64) for m in ensemble]
    floor = {n: max(d[n] for d in dists) for n in err}
    bound = {n: min(max(2.0 * floor[n] + 0.01, 0.04 if np.asarray(G64[n]).size < 8 else 0.0), tol["cap"]) for n in err}
    v, f = np.array(list(err.values())), np.array([floor[n] for n in err])
    b = np.array([bound[n] for n in err])
    power = {d: int((np.sqrt(d * d + f * f) > b).sum()) for d in (0.03, 0.05)}
    worst = sorted(((err[n], floor[n], n) for n in err), reverse=True)[:4]
    print("%s: max |dlogit| %.4g, loss rel %.3g; gradient L2 error HIP-oracle64 median %.4f p90 %.4f max %.4f | floor (ensemble-oracle64) "
          "median %.4f p90 %.4f max %.4f | a systematic 3 %% / 5 %% error would fail %d / %d of %d tensors | worst (err, floor): %s"
          % (label, dl, lrel, np.median(v), np.quantile(v, 0.9), v.max(), np.median(f), np.quantile(f, 0.9), f.max(), power[0.03], power[0.05], v.size,
             [(round(e, 4), round(fl, 4), "/".join(n.split("/")[-3:])) for e, fl, n in worst]))
    assert dl < tol["logit"], (label, dl)
    assert lrel < tol["loss"], (label, lrel)
    bad = [(n, round(err[n], 4), round(bound[n], 4)) for n in err if not err[n] <= bound[n]]
    assert not bad, "%s: gradients beyond 2 x their rounding-cascade floor + 1 %%: %s" % (label, bad)
    assert np.median(v) < tol["grad_median"], (label, float(np.median(v)))
    assert power[0.05] >= min_power * v.size, (label, power)    # the check must be able to see a 5 % defect in most tensors (B = 24 with
    # dropout: a third -- 24 rows under a 0.1 / 0.5 dropout leave few terms per gradient, the family of evaluations spreads wider)
    return err


def _compare_exact(tr, loss, ref, tol, label):
    lref, (c_ref, o_ref, yb_ref), G = ref
    (c, o), yb = tr.last["out"]
    dl = max(float(np.abs(x.detach().float().cpu().numpy() - r).max()) for x, r in ((c, c_ref), (o, o_ref), (yb, yb_ref)))
    lrel = abs(loss - lref) / abs(lref)
    err = _dist(_hip_grads(tr), G)
    v = np.array(list(err.values()))
    print("%s: max |dlogit| %.4g, loss rel %.3g, gradient L2 errors median %.4f p90 %.4f max %.4f" % (label, dl, lrel, np.median(v), np.quantile(v, 0.9), v.max()))
    assert dl < tol["logit"] and lrel < tol["loss"] and v.max() < tol["grad"], (label, dl, lrel, float(v.max()))


@pytest.mark.parametrize("B,lengths,weights,dropout", [(24, "ragged", "random", False), (24, "full", "ones", False), (24, "ragged", "random", True),
                                                       (352, "ragged", "random", False)])
def test_e64_bf16_every_gradient_within_a_few_percent_of_the_storage_rounding_oracle(cuda, monkeypatch, B, lengths, weights, dropout):
    if B * 50 < ops.WGRAD320_MIN_ROWS or lengths == "ragged":
        # (ragged: the sequences run on packed rows, about half of B * 50 -- the wide-block weight-gradient kernel must still be the one that runs)
        monkeypatch.setattr(ops, "WGRAD320_MIN_ROWS", 256 if (lengths == "ragged" and B < 100) else 1024)
    tr, loss, (P, inputs, mask, so, step_seed) = _run(cuda, B, lengths, weights, dropout)
    ref64 = OT.loss_and_grads(P, inputs, mask, so, step_seed=step_seed, storage="bf16")
    ens = _ensemble(P, inputs, mask, so, step_seed)
    _compare_calibrated(tr, loss, ref64, ens, "E64 bf16 B=%d%s vs storage-rounding oracle" % (B, " dropout" if dropout else ""),
                        min_power=0.5 if B >= 64 else 0.3)


def test_the_storage_rounding_oracle_is_the_same_function(cuda):
    """The rounded oracle restates the SAME function: against the exact one its logits move by bf16 rounding only, and with rounding
    switched off the re-associated decoder attention (the form dmt_q1mem computes) equals multihead_attention to 1e-12."""
    sp = S.scaled_spec(S.e64_spec(), E64_ROWS)
    so = dict(sp)
    P = _params(so)
    inputs, mask, _ = make_batch(sp, 9, seed=8, lengths="ragged", weights="random")
    l0, (c0, o0, y0), G0 = OT.loss_and_grads(P, inputs, mask, so)
    l1, (c1, o1, y1), G1 = OT.loss_and_grads(P, inputs, mask, so, storage="bf16")
    assert max(np.abs(c1 - c0).max(), np.abs(o1 - o0).max(), np.abs(y1 - y0).max()) < 3e-2 and abs(l1 - l0) < 1e-2 * abs(l0)
    Pt = OT.to_torch(P)
    rng = np.random.default_rng(1)
    y = torch.tensor(rng.standard_normal((5, 1, 320)))
    mem = torch.tensor(rng.standard_normal((5, 50, 320)))
    lens = torch.tensor([50, 1, 7, 33, 49])
    s = OT._trans_prefix(1) + "num_blocks_0/vanilla_attention/"
    a = OT._mha(y, mem, torch.ones(5, dtype=torch.long), lens, 4, Pt, s)
    b = OT._mha_q1mem(y, mem, lens, 4, Pt, s)
    assert float((a - b).abs().max()) < 1e-12


def test_configs1_at_full_size_forward_loss_and_every_gradient_match_the_oracle(cuda):
    """BASELINE configs[1] as bench.py runs it -- B = 4096, L = 50/50/10, E64, bf16, every default kernel, no threshold lowered --
    against the oracle: ~20 s of oracle time per form on the test box's host cores."""
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    B = 4096
    assert B * 50 >= ops.WGRAD320_MIN_ROWS
    tr, loss, (P, inputs, mask, so, _seed) = _run(cuda, B, "ragged", "random", False, seed=17)
    assert tr.last["out"][0][0].shape[0] == B
    # (a) the storage-rounding oracle, bound calibrated per tensor
    ref_r = OT.loss_and_grads(P, inputs, mask, so, storage="bf16")
    ens = _ensemble(P, inputs, mask, so, None)
    _compare_calibrated(tr, loss, ref_r, ens, "configs[1] B=4096 vs storage-rounding oracle")
    # (b) the exact oracle: the accuracy of the bf16 mode itself at full size (the bounds of tests/test_gpu_e64.py)
    ref_x = OT.loss_and_grads(P, inputs, mask, so)
    _compare_exact(tr, loss, ref_x, dict(logit=6e-2, loss=3e-2, grad=0.2), "configs[1] B=4096 vs exact oracle")
    # sampled embedding rows, element-wise: the rows of the largest table that the batch touched most and least
    g_sku = sparse_to_dense_tables(tr.store, tr.engine.sparse)["embedding_trans/Sku/embedding"]
    r_sku = np.asarray(ref_r[2]["embedding_trans/Sku/embedding"])
    norms = np.linalg.norm(r_sku, axis=1)
    touched = np.nonzero(norms > 0)[0]
    assert touched.size > 100 and np.all(np.abs(g_sku[norms == 0]) == 0)             # untouched rows get exactly nothing
    order = touched[np.argsort(norms[touched])]
    for r in list(order[-8:]) + list(order[:8]):
        assert np.linalg.norm(g_sku[r] - r_sku[r]) <= 0.08 * norms[r] + 1e-3 * norms.max(), (int(r), float(norms[r]))

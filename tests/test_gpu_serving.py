"""Serving path (saved_model/export_model.py:23-138, model/inference_mlp.py:73-113): encode the user's
sequences once, decode per candidate -- same logits as the predict graph on the tiled batch, and as the oracle."""
import numpy as np
import pytest
import torch

from oracle import dmt_oracle as O
from cikm2020_dmt_amd import spec as S
from cikm2020_dmt_amd.data_feed.synthetic import make_batch
from cikm2020_dmt_amd.serving import CandidateScorer, normalisation_constants, normalise_dense
from cikm2020_dmt_amd.train import Trainer
from tests.util import small_specs

pytestmark = pytest.mark.gpu


def _tile_user_side(sp, inputs):
    """What inference_mlp.online_build_sparsetensor does: every user-side ('u') feature of row 0 repeated for all rows."""
    from cikm2020_dmt_amd.sparse import SparseTensorValue
    B = inputs["features"].shape[0]
    out = dict(inputs)
    for (_n, _r, _d, f, side) in sp["embedding_list"]:
        if side != "u":
            continue
        for key in (f, f + "Wts"):
            spv = inputs[key]
            rows = np.asarray(spv.indices)[:, 0]
            sel = rows == 0
            cols = np.asarray(spv.indices)[sel, 1]
            vals = np.asarray(spv.values)[sel]
            idx = np.stack([np.repeat(np.arange(B), len(cols)), np.tile(cols, B)], axis=1).astype(np.int64)
            out[key] = SparseTensorValue(idx, np.tile(vals, B), (B, spv.dense_shape[1]))
    return out


@pytest.mark.parametrize("dtype,tol,variant", [(torch.float32, 2e-5, None), (torch.bfloat16, 4e-2, None), (torch.float32, 2e-5, "sin_cos"), (torch.float32, 2e-5, "in_mlp")])
def test_encode_once_equals_tiled_predict_graph_and_oracle(cuda, dtype, tol, variant):
    so, sp = small_specs()
    if variant == "sin_cos":      # position_sin_cos + is_decoder_add_pos_emb + is_trans_out_concat_item: the request path adds the same constants as the train graph
        opt = dict(position_encoding_method="position_sin_cos", is_decoder_add_pos_emb=True, is_trans_out_concat_item=True)
        so, sp = dict(so, **opt), dict(sp, **opt)
    if variant == "in_mlp":       # is_trans_input_by_mlp + two encoder blocks: the request path runs the same input layers and prep, once per request
        opt = dict(is_trans_input_by_mlp=True, num_blocks_encode=2, num_blocks_decode=2, is_trans_out_concat_item=True)
        so, sp = dict(so, **opt), dict(sp, **opt)
    P = O.init_params(so, seed=21)
    inputs, mask, label = make_batch(sp, 37, seed=8, lengths="ragged", weights="random")
    tiled = _tile_user_side(sp, inputs)
    tr = Trainer(sp, device=cuda, compute_dtype=dtype, init=False, dropout=False)
    tr.store.load_state(P)
    batch = tr.make_batch(tiled, mask, label)
    with torch.no_grad():
        c_ref, o_ref = tr.engine.inference(batch, is_predict=True)
    sc = CandidateScorer(tr.engine, export_weight=(1.0, 3.0))
    c, o = sc.logits(batch)
    assert torch.allclose(c.float(), c_ref.float(), atol=tol, rtol=tol)
    assert torch.allclose(o.float(), o_ref.float(), atol=tol, rtol=tol)
    # ... and the oracle's predict graph on the same tiled inputs
    c_or, o_or = O.inference(tiled, P, so, is_predict=True)
    assert np.abs(c.float().cpu().numpy() - c_or).max() < (3e-4 if dtype == torch.float32 else 8e-2)
    score, pc, po = sc.score(batch)
    want = O.serving_scores(c_or, o_or, (1.0, 3.0))
    assert np.abs(score.cpu().numpy() - want).max() < (1e-4 if dtype == torch.float32 else 2e-2)


def test_request_assembly_and_dense_normalisation(cuda):
    so, sp = small_specs()
    P = O.init_params(so, seed=5)
    B, F = 19, sp["feature_dimension"]
    rng = np.random.default_rng(2)
    mean, std = rng.uniform(0.0, 4.0, F), rng.uniform(0.05, 3.0, F)
    raw = rng.uniform(-1.0, 30.0, (B, F)).astype(np.float32)
    want = O.serving_normalise(raw, mean, std)
    c, s = normalisation_constants(mean, std)
    got = normalise_dense(torch.as_tensor(raw).to(cuda), torch.as_tensor(c).to(cuda), torch.as_tensor(s).to(cuda)).cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=2e-6)
    assert got.min() >= -0.99 and got.max() <= 0.99 and (np.abs(got) < 0.99).any()

    tr = Trainer(sp, device=cuda, compute_dtype=torch.float32, init=False, dropout=False)
    tr.store.load_state(P)
    inputs, mask, label = make_batch(sp, B, seed=3, lengths="ragged")
    tiled = _tile_user_side(sp, inputs)
    tiled["features"] = want                                     # the tiled predict graph sees the normalised features
    ref_batch = tr.make_batch(tiled, mask, label)
    user_inputs, item_inputs = {}, {}
    for (_n, _r, _d, f, side) in sp["embedding_list"]:
        col = ref_batch.feats[f]
        idx, lens = col.idx.cpu().numpy(), col.lens.cpu().numpy()
        wts = col.wts.cpu().numpy() if col.wts is not None else None
        if side == "u":
            user_inputs[f] = (idx[0, :lens[0]], wts[0, :lens[0]] if wts is not None else None)
        else:
            item_inputs[f] = (idx, lens, wts)
    for (_n, _r, _d, f, _side) in sp["embedding_list_bias"]:     # bias-tower features are not part of the predict graph
        col = ref_batch.feats[f]
        item_inputs.setdefault(f, (col.idx.cpu().numpy(), col.lens.cpu().numpy(), None))
    sc = CandidateScorer(tr.engine, mean=mean, std=std)
    req = sc.tile_request(user_inputs, item_inputs, raw)
    c, o = sc.logits(req)
    with torch.no_grad():
        c_ref, o_ref = tr.engine.inference(ref_batch, is_predict=True)
    assert torch.allclose(c, c_ref, atol=2e-5, rtol=2e-5) and torch.allclose(o, o_ref, atol=2e-5, rtol=2e-5)


def test_hip_graph_replay_matches_eager_scoring(cuda):
    so, sp = small_specs()
    P = O.init_params(so, seed=33)
    tr = Trainer(sp, device=cuda, compute_dtype=torch.bfloat16, init=False, dropout=False)
    tr.store.load_state(P)
    sc = CandidateScorer(tr.engine, export_weight=(2.0, 1.0))
    from cikm2020_dmt_amd.serving import GraphedScorer

    def request(seed):
        inputs, mask, label = make_batch(sp, 64, seed=seed, lengths="full")
        return tr.make_batch(_tile_user_side(sp, inputs), mask, label)

    gs = GraphedScorer(sc, request(1))
    for seed in (2, 3):
        req = request(seed)
        want = [t.clone() for t in sc.score(req)]
        got = gs.score(req)
        for a, b in zip(got, want):
            assert torch.equal(a, b)

"""Serving path: score B candidate items for ONE user.

Mirrors /root/reference/DMT_code/saved_model/export_model.py:23-138 (online normalisation of the raw dense features,
sigmoid heads, weighted score) and model/inference_mlp.py:73-113 (`online_build_sparsetensor`: the user-side id features of
the request are tiled across the candidate batch and the ordinary predict graph runs on the tiled batch).

The reference re-encodes the user's behaviour sequences for every candidate row.  Here the three sequence encoders (self-
attention + FFN over [T, d]) and the K | V projections of their memories run ONCE per request; every candidate only pays for
its own embedding lookups, the single-query decoder attention against the shared memory (zero batch stride), MMoE and the
towers.  Same numbers as Inference.inference(tiled batch, is_predict=True) -- checked in tests/test_gpu_serving.py.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import numpy as np
import torch

from .engine import AssembleFn, DeviceBatch, DMTEngine, FeatureColumn


def normalisation_constants(mean: Sequence[float], std: Sequence[float]):
    """saved_model/preprocess.py:17-40 `vec_constant` (float64 there as well):
    c = mean*std / (3 (std+eps)^2) + mean*std / (std+eps) - mean."""
    m, s = np.asarray(mean, dtype=np.float64), np.asarray(std, dtype=np.float64)
    eps = 1e-7
    c = m * s / (np.square(s + eps) * 3.0) + m * s / (s + eps) - m
    return c.astype(np.float32), s.astype(np.float32)


def normalise_dense(raw: torch.Tensor, const_vec: torch.Tensor, std: torch.Tensor) -> torch.Tensor:
    """export_model.py:86-96: clip(x, 0, FLT_MAX) * std / (3 (std + 1e-7)^2) - c, clipped to [-0.99, 0.99] (fp32)."""
    x = torch.clamp(raw.float(), min=0.0)      # (the upper bound sys.float_info.max is no bound for float32)
    y = x * std / (torch.square(std + 1e-7) * 3.0) - const_vec
    return torch.clamp(y, min=-0.99, max=0.99)


class CandidateScorer:
    def __init__(self, engine: DMTEngine, export_weight=(1.0, 1.0), mean: Optional[Sequence[float]] = None,
                 std: Optional[Sequence[float]] = None, optimizer=None):
        """optimizer: the TFAdam of a LIVE trainer sharing this engine's tables, or None for a frozen / restored model.  The exact lazy
        Adam leaves zero-gradient updates of untouched rows pending; scoring reads rows without replaying them, so with
        a live optimizer the tables are flushed before every request that follows a train step."""
        self.engine = engine
        self.optimizer = optimizer
        self._flushed_at = -1
        self.spec = engine.spec
        self.w = (float(export_weight[0]), float(export_weight[1]))
        dev = engine.store.device
        if mean is not None:
            c, s = normalisation_constants(mean, std)
            self.const_vec, self.std = torch.as_tensor(c).to(dev), torch.as_tensor(s).to(dev)
        else:
            self.const_vec = self.std = None
        self.user_feats = [f for (_n, _r, _d, f, side) in self.spec["embedding_list"] if side == "u"]

    # ------------------------------------------------------------------ request assembly
    def tile_request(self, user_inputs: dict, item_inputs: dict, dense_raw: np.ndarray) -> DeviceBatch:
        """user_inputs: {feature: (ids int[T], wts float[T] or None)} -- one list per user-side feature, as the
        emb_common_* placeholders of export_model.py:64-84;  item_inputs: {feature: (idx int[B, T], lens int[B], wts or None)}
        (already padded; item-side features carry one id per candidate in the reference data);  dense_raw: float[B, F]."""
        dev = self.engine.store.device
        B = int(np.asarray(dense_raw).shape[0])
        feats: Dict[str, FeatureColumn] = {}
        for f, (ids, wts) in user_inputs.items():
            ids = np.asarray(ids, dtype=np.int32).reshape(1, -1)
            T = max(ids.shape[1], 1)
            idx = torch.as_tensor(ids).to(dev).expand(B, T).contiguous()
            lens = torch.full((B,), ids.shape[1], dtype=torch.int32, device=dev)
            w = None
            if wts is not None and not np.all(np.asarray(wts) == 1.0):
                w = torch.as_tensor(np.asarray(wts, dtype=np.float32).reshape(1, -1)).to(dev).expand(B, T).contiguous()
            feats[f] = FeatureColumn(idx, w, lens, T)
        for f, (idx, lens, wts) in item_inputs.items():
            idx = np.asarray(idx, dtype=np.int32)
            w = None
            if wts is not None and not np.all(np.asarray(wts) == 1.0):
                w = torch.as_tensor(np.asarray(wts, dtype=np.float32)).to(dev)
            feats[f] = FeatureColumn(torch.as_tensor(idx).to(dev), w, torch.as_tensor(np.asarray(lens, dtype=np.int32)).to(dev), idx.shape[1])
        dense = torch.as_tensor(np.asarray(dense_raw, dtype=np.float32)).to(dev)
        if self.const_vec is not None:
            dense = normalise_dense(dense, self.const_vec, self.std)
        return DeviceBatch(B, feats, dense.contiguous())

    @staticmethod
    def _row0(batch: DeviceBatch) -> DeviceBatch:
        feats = {f: FeatureColumn(c.idx[:1], c.wts[:1] if c.wts is not None else None, c.lens[:1], c.T) for f, c in batch.feats.items()}
        return DeviceBatch(1, feats, batch.dense[:1])

    # ------------------------------------------------------------------ scoring
    @torch.no_grad()
    def logits(self, batch: DeviceBatch):
        """(click_logit, order_logit) [B, 1] for B candidate rows of one user; user-side columns are read from row 0."""
        eng, spec = self.engine, self.spec
        if self.optimizer is not None and self.optimizer.global_step != self._flushed_at:
            self.optimizer.flush_tables()
            self._flushed_at = self.optimizer.global_step
        saved_seed, eng.dropout_step_seed = eng.dropout_step_seed, None        # predict graph: is_train=False
        try:
            b1 = self._row0(batch)
            X1, _tar1, _z1 = eng.gather(b1)
            tar_g, zbuf = eng.gather_pooled(batch)
            in_mlp = bool(spec.get("is_trans_input_by_mlp"))      # (then both gathers hand out raw rows and every sequence has its own target)
            us, tars_scaled = [], []
            for i, pairs in enumerate(spec["attention_embed_pairs"]):
                lens1 = b1.feats[pairs[-1][0]].lens
                x1, tar_i = eng.input_mlp(i, X1[i], tar_g) if in_mlp else (X1[i], tar_g)
                tars_scaled.append(tar_i)
                mem1 = eng.encode_prepared(x1, lens1, i)                       # [1, T, d], once per request
                k_lens = lens1.expand(batch.B).contiguous()
                us.append(eng.decode_shared(eng.decoder_query(tar_i).unsqueeze(1), mem1, k_lens, i).squeeze(1))
            z = AssembleFn.apply(zbuf, eng.plan.interest_off, spec["d_model"], *eng.interest_blocks(us, tars_scaled))
            tasks = eng.expert_gate(z, z_is_engine_buffer=True)          # (z is gather_pooled's zero-initialised zbuf)
            return tuple(eng.build_tower(m, nm) for m, nm in zip(tasks, ("click", "order")))
        finally:
            eng.dropout_step_seed = saved_seed

    @torch.no_grad()
    def score(self, batch: DeviceBatch):
        """export_model.py:106-114: Scores = (w0 sigmoid(click) + w1 sigmoid(order)) / (w0 + w1); also returns both heads."""
        c, o = self.logits(batch)
        pc, po = torch.sigmoid(c.float()).reshape(-1), torch.sigmoid(o.float()).reshape(-1)
        return (self.w[0] * pc + self.w[1] * po) / (self.w[0] + self.w[1]), pc, po


class GraphedScorer:
    """The kernel sequence of CandidateScorer.score for ONE request shape (candidate count, padded id-list lengths), captured
    once into a HIP graph and replayed per request: the eager path is bound by ~100 host-side launches (1.3 ms at any candidate
    count), the replay is not (0.63 ms at 256 candidates, 0.89 ms at 4096 on MI355X, scripts/serve_bench.py).
    Requests are written into the captured input buffers in place; shapes must match the template."""

    def __init__(self, scorer: CandidateScorer, template: DeviceBatch):
        self.scorer = scorer
        self.static = template
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                 # warm-up outside the capture (lazy library / allocator state)
            for _ in range(2):
                scorer.score(template)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = scorer.score(template)

    def score(self, batch: DeviceBatch):
        st = self.static
        if batch.B != st.B or set(batch.feats) != set(st.feats):
            raise ValueError("request does not match the captured shape (B=%d)" % st.B)
        for f, col in batch.feats.items():
            dst = st.feats[f]
            if col.idx.shape != dst.idx.shape:
                raise ValueError("feature %s: id-list shape differs from the captured request" % f)
            # whether a weights column exists is a property of the captured graph, not of a request's values: a request whose weights
            # are all 1.0 arrives without the column (tile_request drops it) and is served by writing ones
            if dst.wts is None and col.wts is not None:
                raise ValueError("feature %s: the captured request has no weights column but this request has weights != 1" % f)
            dst.idx.copy_(col.idx); dst.lens.copy_(col.lens)
            if dst.wts is not None:
                if col.wts is not None:
                    dst.wts.copy_(col.wts)
                else:
                    dst.wts.fill_(1.0)
        st.dense.copy_(batch.dense)
        self.graph.replay()
        return self.out

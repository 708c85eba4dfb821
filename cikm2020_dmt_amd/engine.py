"""Execution engine of the DMT hot path on one MI355X: batches in HBM, gather plan, forward graph.

The reference-shaped classes under cikm2020_dmt_amd/model/ are thin facades over this module; every
compute step is a libdmt_hip.so kernel reached through cikm2020_dmt_amd/ops.py.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib as L
from . import ops
from .ops import F32, Weight
from .spec import mmoe_input_width, trans_prefix
from .variables import VariableStore


# ------------------------------------------------------------------------------------------------ batch
class FeatureColumn:
    __slots__ = ("idx", "wts", "lens", "T", "lens_host")

    def __init__(self, idx, wts, lens, T, lens_host=None):
        self.idx, self.wts, self.lens, self.T = idx, wts, lens, T
        self.lens_host = lens_host          # numpy int32 [B] (the lengths as the host saw them) or None: no packed rows for this column


class SeqPack:
    """Packed-row layout of ONE behaviour sequence of a batch (include/dmt_hip.h, "PACKED ROWS"): example b's rows t < len[b] are rows
    row_off[b] + t of [R, d] matrices, R = sum len.  Examples are grouped by length class (len <= 16 / <= 32 / <= 64 -- the padded lengths
    Tp the fused self-attention kernel tiles with), so a class is a contiguous range of `order` and of the rows.  Built on the host from
    the batch's lengths (a few vectorised numpy passes over B entries), uploaded once as one int32 buffer and cached on the batch.
    The reference computes every sequence at [B, T_max] (SURVEY.md F13 notes that a new kernel may skip the padded rows): on its
    data -- click histories of 1 - 50 items -- about half of those rows are padding."""

    def __init__(self, lens_np, T, device):
        B = int(lens_np.shape[0])
        ln = lens_np.astype(np.int64)
        self.B, self.T = B, int(T)
        cls = np.where(ln <= 16, 0, np.where(ln <= 32, 1, 2))
        order = np.argsort(cls, kind="stable")
        ln_s = ln[order]
        off_s = np.zeros(B, np.int64)
        off_s[1:] = np.cumsum(ln_s)[:-1]
        self.R = int(ln_s.sum())
        row_off = np.empty(B, np.int64)
        row_off[order] = off_s
        self.n_cls = [int((cls == c).sum()) for c in range(3)]
        tabs, pos = [], 0
        for c in range(3):
            n, lg = self.n_cls[c], 4 + c
            if n == 0:
                continue
            epw = 256 >> lg                                   # examples per 256-row tile
            nt = (n + epw - 1) // epw
            ent = np.zeros((nt * epw, 4), np.int32)
            ent[:, 0], ent[:, 3] = -1, lg
            ex = order[pos: pos + n]
            ent[:n, 0], ent[:n, 1], ent[:n, 2] = ex, ln[ex], row_off[ex]
            if lg == 4:                                       # two examples per 32-row block
                tab = ent.reshape(nt, 8, 2, 4)
            elif lg == 5:                                     # one example per block
                tab = np.repeat(ent.reshape(nt, 8, 1, 4), 2, axis=2)
            else:                                             # one example spans two blocks
                tab = np.repeat(np.repeat(ent.reshape(nt, 4, 1, 1, 4), 2, axis=2), 2, axis=3).reshape(nt, 8, 2, 4)
            tabs.append(tab)
            pos += n
        blocks = np.concatenate(tabs, axis=0) if tabs else np.zeros((0, 8, 2, 4), np.int32)
        self.n_tiles = int(blocks.shape[0])
        # one upload: [blocks | row_off | order]  (blocks first: the kernel reads 16-byte entries)
        nb = blocks.size
        buf = np.empty(nb + 2 * B, np.int32)
        buf[:nb] = blocks.reshape(-1)
        buf[nb: nb + B] = row_off
        buf[nb + B:] = order
        dbuf = torch.from_numpy(buf).to(device)
        self.buf = dbuf
        self.blocks, self.row_off, self.order = dbuf[:nb], dbuf[nb: nb + B], dbuf[nb + B:]
        self.n_short = self.n_cls[0] + self.n_cls[1]         # examples of at most 32 rows: order[:n_short]

    @staticmethod
    def eligible(lens_np, T):
        """Packed rows are defined for lengths in [1, T] (a length of 0 means "attend uniformly over padding" in the dense layout:
        SURVEY.md Appendix A.1 -- the data never has it)."""
        return lens_np is not None and lens_np.size > 0 and int(lens_np.min()) >= 1 and int(lens_np.max()) <= T


class DeviceBatch:
    """One batch at the post-vocabulary-lookup boundary, resident in HBM as padded int32 columns."""

    def __init__(self, B, feats: Dict[str, FeatureColumn], dense, mask=None, label=None):
        self.B, self.feats, self.dense, self.mask, self.label = B, feats, dense, mask, label
        self._prep = None
        self._packs = {}             # sequence -> SeqPack or None (DMTEngine.seq_pack)
        # "the upload of this batch is done": what the index-plane stream of the Trainer waits for instead of the whole compute stream
        self.ready = None
        if dense.is_cuda:
            self.ready = torch.cuda.Event()
            self.ready.record(ops.cur_stream(dense.device))

    @staticmethod
    def from_inputs(inputs: dict, spec: dict, device, mask=None, label=None, pad_to: Optional[Dict[str, int]] = None) -> "DeviceBatch":
        """inputs: {'features': float[B,F], f: SparseTensorValue(int ids), f+'Wts': SparseTensorValue(float)} --
        what tfrecord_mask.parse_single_line + LookupTables.transform_id2index hand to Inference.inference."""
        dense = torch.as_tensor(np.ascontiguousarray(inputs["features"], dtype=np.float32)).to(device)
        B = dense.shape[0]
        names = [f for (_n, _r, _d, f, _s) in spec["embedding_list"]] + [f for (_n, _r, _d, f, _s) in spec["embedding_list_bias"]]
        feats = {}
        for f in names:
            if f in feats:
                continue
            sp = inputs[f]
            T = sp.dense_shape[1]
            if pad_to and f in pad_to:
                T = max(T, pad_to[f])
            T = max(T, 1)
            idx_np, lens_np = sp.to_padded(T)
            wts = None
            wsp = inputs.get(f + "Wts")
            if wsp is not None and len(wsp.values) and not np.all(np.asarray(wsp.values) == 1.0):
                w_np, _ = wsp.to_padded(T)
                wts = torch.as_tensor(w_np.astype(np.float32)).to(device)
            feats[f] = FeatureColumn(torch.as_tensor(idx_np.astype(np.int32)).to(device), wts,
                                     torch.as_tensor(lens_np.astype(np.int32)).to(device), T, lens_host=np.asarray(lens_np, dtype=np.int32))
        m = torch.as_tensor(np.asarray(mask, dtype=np.float32)).to(device) if mask is not None else None
        lb = torch.as_tensor(np.asarray(label, dtype=np.float32)).to(device) if label is not None else None
        return DeviceBatch(B, feats, dense, m, lb)

    @staticmethod
    def from_columns(cols: dict, spec: dict, device) -> "DeviceBatch":
        """cols: the output of data_feed.native.BatchParser.parse (libdmt_input.so): per id feature f  f -> int32 [B,T],
        f+'Wts' -> float32 [B,T], f+'/lens' -> int32 [B];  'features' [B,F], 'mask' [B,5], 'label' [B,1].
        Same batch as from_inputs(..., pad_to=T): the weights column is dropped when every weight is 1 (unweighted mean)."""
        names = [f for (_n, _r, _d, f, _s) in spec["embedding_list"]] + [f for (_n, _r, _d, f, _s) in spec["embedding_list_bias"]]
        packed = cols.get("__buffer__")
        if packed is not None and isinstance(packed[0], torch.Tensor):
            # one (asynchronous, page-locked) upload of the whole batch; the columns are views of the device copy
            hbuf, layout, id_feats = packed
            dbuf = hbuf.to(device, non_blocking=True)
            tdt = {np.dtype(np.int32): torch.int32, np.dtype(np.float32): torch.float32}

            def view(name):
                o, dt, shape = layout[name]
                n = int(np.prod(shape)) * np.dtype(dt).itemsize
                return dbuf[o:o + n].view(tdt[np.dtype(dt)]).view(*shape)

            not_one = cols["__wts_not_one__"]
            feats = {}
            for f in dict.fromkeys(names):
                T = layout[f][2][1]
                wts = view(f + "Wts") if int(not_one[id_feats.index(f)]) != 0 else None
                o_l, dt_l, shape_l = layout[f + "/lens"]
                lens_h = hbuf[o_l: o_l + int(np.prod(shape_l)) * 4].view(torch.int32).numpy().copy()     # (the pinned buffer is reused by the parser)
                feats[f] = FeatureColumn(view(f), wts, view(f + "/lens"), T, lens_host=lens_h)
            dense = view("features")
            m = view("mask") if "mask" in layout else None
            lb = view("label")[:, 0] if "label" in layout else None
            slot = cols.get("__ring_slot__")
            if slot is not None and dbuf.is_cuda:
                # (BatchParser.ring: the host buffer is reused `ring` batches from now -- not before this copy has read it)
                ev = torch.cuda.Event()
                ev.record(ops.cur_stream(dbuf.device))
                slot["event"] = ev
            return DeviceBatch(dense.shape[0], feats, dense, m, lb)
        dense = torch.as_tensor(cols["features"]).to(device, non_blocking=True)
        B = dense.shape[0]
        feats = {}
        for f in dict.fromkeys(names):
            idx, lens, w = cols[f], cols[f + "/lens"], cols.get(f + "Wts")
            T = idx.shape[1]
            wts = None
            if w is not None:
                valid = np.arange(T)[None, :] < lens[:, None]
                if valid.any() and not np.all(w[valid] == 1.0):
                    wts = torch.as_tensor(w).to(device, non_blocking=True)
            feats[f] = FeatureColumn(torch.as_tensor(idx).to(device, non_blocking=True), wts, torch.as_tensor(lens).to(device, non_blocking=True), T,
                                     lens_host=np.array(lens, dtype=np.int32))
        m = torch.as_tensor(cols["mask"]).to(device, non_blocking=True) if "mask" in cols else None
        lb = torch.as_tensor(cols["label"][:, 0]).to(device, non_blocking=True) if "label" in cols else None
        return DeviceBatch(B, feats, dense, m, lb)


# ------------------------------------------------------------------------------------------------ gather
class _GatherPlan:
    """Static description of every (table, id column) pair: output offsets, groups, global rows."""

    def __init__(self, spec: dict, store: VariableStore):
        self.spec = spec
        d = spec["d_model"]
        self.items = []          # dicts: feature, table tf_name, dim, rows, pooled_off, seq_id, seq_off, group
        pooled_off = spec["feature_dimension"]
        seq_of, off_in_seq = {}, {}
        for s, pairs in enumerate(spec["attention_embed_pairs"]):
            col = 0
            dim_of = {f: dm for (_n, _r, dm, f, _s) in spec["embedding_list"]}
            for (uf, itf) in pairs:
                seq_of[uf], off_in_seq[uf] = s, col
                seq_of[itf], off_in_seq[itf] = L.DMT_SEQ_TARGET, col
                col += dim_of[uf]
            if col != d:
                raise ValueError("sequence %d field widths sum to %d, d_model is %d" % (s, col, d))
        group_of = {}
        for s, pairs in enumerate(spec["attention_embed_pairs"]):
            for (uf, _itf) in pairs:
                group_of[uf] = 1 + s
            if spec["attention_embed_seq_ts"]:
                group_of[spec["attention_embed_seq_ts"][s]] = 1 + s
        for (name, rows, dim, feat, _side) in spec["embedding_list"]:
            self.items.append(dict(feature=feat, table="embedding_trans/%s/embedding" % name, dim=dim, rows=rows,
                                   pooled_off=pooled_off, seq_id=seq_of.get(feat, -1), seq_off=off_in_seq.get(feat, 0),
                                   group=group_of.get(feat, 0)))
            pooled_off += dim
        self.K = mmoe_input_width(spec)
        self.interest_off = pooled_off
        self.interest_blocks = len(spec["attention_embed_pairs"]) * (2 if (spec.get("is_trans_out_concat_item")
                                                                           and not spec.get("is_trans_out_by_mlp")) else 1)
        assert self.interest_off + self.interest_blocks * d == self.K
        self.bias_off = (self.K + 3) // 4 * 4
        boff = self.bias_off
        for (name, rows, dim, feat, _side) in spec["embedding_list_bias"]:
            self.items.append(dict(feature=feat, table="%s/embedding" % name, dim=dim, rows=rows, pooled_off=boff, seq_id=-1,
                                   seq_off=0, group=50))
            boff += dim
        self.bias_width = boff - self.bias_off
        self.ldz = (boff + 7) // 8 * 8
        self.max_dim = max(it["dim"] for it in self.items)
        if len(self.items) > L.DMT_MAX_FEATURES:
            raise ValueError("too many embedding features (%d > %d)" % (len(self.items), L.DMT_MAX_FEATURES))
        for it in self.items:
            it["row_base"] = store.table_rows[it["table"]][0]


class FanOutFn(torch.autograd.Function):
    """x -> n aliases of x, ONE backward node that sums their gradients.  The target item's embedding feeds all three decoders: left
    to autograd, the three gradients meet in the input buffer of the gather node, and the engine queues "wait for lane C's decoder
    backward, add" on the compute stream BEFORE it gets to issue sequence 0's backward there (it issues nodes newest-first): the
    lanes then start one after the other's decoder instead of together.  This node is older than every sequence, so its waits are
    queued after all three backward pipelines (measured: 9.75 -> 9.61 ms/step)."""

    @staticmethod
    def forward(ctx, x, n):
        return tuple(x.view_as(x) for _ in range(n))

    @staticmethod
    def backward(ctx, *gs):
        gs = [g for g in gs if g is not None]
        if not gs:
            return None, None
        acc = gs[0]
        for g in gs[1:]:
            acc = acc + g
        return acc, None


class GatherFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, engine, batch, packs, *pos_leaves):
        plan, store, spec = engine.plan, engine.store, engine.spec
        dev, cdt = store.device, store.compute_dtype
        B, d = batch.B, spec["d_model"]
        n_seq = len(spec["attention_embed_pairs"])
        seq_T = []
        for pairs in spec["attention_embed_pairs"]:
            seq_T.append(max(batch.feats[uf].T for (uf, _i) in pairs))
        # packs[s]: SeqPack -> the sequence's rows are produced packed, [1, R, d] (one "example" of R rows for every row-wise op)
        X = [torch.empty((1, max(packs[s].R, 1), d) if packs[s] is not None else (B, seq_T[s], d), dtype=cdt, device=dev) for s in range(n_seq)]
        tar = torch.empty((B, d), dtype=cdt, device=dev)
        zbuf = torch.zeros((B, plan.ldz), dtype=cdt, device=dev)
        desc = L.GatherDesc()
        desc.B, desc.n_features = B, len(plan.items)
        inv = torch.empty((len(plan.items), B), dtype=F32, device=dev)
        cache = engine.row_cache(batch) if store.shard is not None else None
        for i, it in enumerate(plan.items):
            col = batch.feats[it["feature"]]
            f = desc.feat[i]
            if cache is None:
                f.table = store.table[it["table"]].data_ptr()
                f.rows, f.dim = it["rows"], it["dim"]
                f.idx = col.idx.data_ptr()
            else:
                # row-sharded tables: the rows this batch reads were fetched from their owners into `cache` (Trainer.sync_rows);
                # ids become cache slots (engine.fetch_rows)
                rows_c, slots, ebase = cache
                f.table = rows_c.data_ptr()
                f.rows, f.dim, f.row_stride = rows_c.shape[0], it["dim"], rows_c.shape[1]
                per = batch.B * col.T
                e0 = ebase[i]
                has_pool, has_seq = it["pooled_off"] >= 0, it["seq_id"] >= 0
                f.idx = slots[e0: e0 + per].data_ptr()
                if has_seq:
                    es = e0 + (per if has_pool else 0)
                    f.idx_seq = slots[es: es + per].data_ptr()
                    if not has_pool:
                        f.idx = f.idx_seq
            f.wts = col.wts.data_ptr() if col.wts is not None else None
            f.lens = col.lens.data_ptr()
            f.T = col.T
            f.pooled_off, f.seq_id, f.seq_off, f.group = it["pooled_off"], it["seq_id"], it["seq_off"], it["group"]
            f.inv_wsum = inv[i].data_ptr()
        desc.n_seq = n_seq
        # is_trans_input_by_mlp (mmoe_transformer_unbias.py:196-198): the dense layers sit between the lookup and the scale / position /
        # dropout prep, so the gather hands out the RAW rows (scale 1, no positions, no dropout) and the engine does the prep afterwards
        raw = bool(spec.get("is_trans_input_by_mlp"))
        for s in range(n_seq):
            desc.seq_out[s] = X[s].data_ptr()
            desc.seq_T[s] = seq_T[s]
            desc.pos[s] = None if raw else pos_leaves[s].data_ptr()
            if packs[s] is not None:
                desc.seq_row_off[s] = packs[s].row_off.data_ptr()
                desc.seq_row_len[s] = engine.seq_lens(batch, s).data_ptr()
            if seq_T[s] > pos_leaves[s].shape[0]:
                raise ValueError("sequence %d length %d exceeds the learned position table (%d rows)" % (s, seq_T[s], pos_leaves[s].shape[0]))
        desc.tar_out = tar.data_ptr()
        desc.d_model = d
        desc.seq_scale = 1.0 if raw else float(d) ** 0.5
        desc.pooled, desc.ld_pooled = zbuf.data_ptr(), plan.ldz
        desc.dense, desc.n_dense = batch.dense.data_ptr(), spec["feature_dimension"]
        desc.out_dtype = ops.dt_code(cdt)
        # block-input dropout of every sequence (TransformerModel.py:101) is applied by the gather itself; its gradient by the
        # consumers of dX (dmt_embgrad_reduce, dmt_colsum_drop) -- same counter mask as ops.dropout(x, rate, seed, 10*s)
        seeds, keep = engine._input_dropout(n_seq) if not raw else ([0] * n_seq, 0.0)
        for s in range(n_seq):
            desc.seq_drop_seed[s] = seeds[s]
        desc.seq_drop_keep = keep
        ctx.drop = (seeds, keep)
        with ops._Timed("gather_fwd", engine.gather_bytes(batch, seq_T)):
            L.call("dmt_gather_fwd", C.byref(desc), ops.stream_ptr())
        ctx.engine, ctx.batch, ctx.inv, ctx.seq_T, ctx.packs = engine, batch, inv, seq_T, packs
        ctx.pos_shapes = [tuple(pl.shape) for pl in pos_leaves]
        ctx.pos_leaves = pos_leaves
        ctx.raw = raw
        return (*X, tar, zbuf)

    @staticmethod
    def backward(ctx, *grads):
        engine, batch = ctx.engine, ctx.batch
        spec = engine.spec
        n_seq = len(spec["attention_embed_pairs"])
        if engine.defer_sparse and all(g is not None for g in grads) and all(ops._grad_view(pl) is not None or not pl.requires_grad or ctx.raw
                                                                              for pl in ctx.pos_leaves):
            # Trainer.train_step finishes this node itself (finish_sparse_backward), on the index lane, beside the deferred weight gradients
            engine._pending_sparse = (ctx, grads)
            return (None, None, None) + (None,) * len(ctx.pos_leaves)
        return GatherFn._finish(ctx, grads)

    @staticmethod
    def _finish(ctx, grads):
        engine, batch = ctx.engine, ctx.batch
        spec = engine.spec
        n_seq = len(spec["attention_embed_pairs"])
        dX = [g.contiguous() if g is not None else None for g in grads[:n_seq]]
        dtar, dz = grads[n_seq], grads[n_seq + 1]
        B, d = batch.B, spec["d_model"]
        dev, cdt = engine.store.device, engine.store.compute_dtype
        packs = ctx.packs
        for s in range(n_seq):
            if dX[s] is None:
                dX[s] = torch.zeros((1, max(packs[s].R, 1), d) if packs[s] is not None else (B, ctx.seq_T[s], d), dtype=cdt, device=dev)
        dtar = dtar.contiguous() if dtar is not None else torch.zeros((B, d), dtype=cdt, device=dev)
        dz = dz.contiguous() if dz is not None else torch.zeros((B, engine.plan.ldz), dtype=cdt, device=dev)
        # learned positions: dP[t] = sum_b dX[b, t]   (lookup by range(T), TransformerModel_util.py:296-306)
        dpos = []
        for s in range(n_seq):
            if ctx.raw or not ctx.pos_leaves[s].requires_grad:      # position_sin_cos: a constant table; raw rows: the prep op has the gradient
                dpos.append(None)
                continue
            # accumulate straight into the position table's gradient arena when it is reachable (no zero-fill, no autograd add)
            gv = ops._grad_view(ctx.pos_leaves[s])
            direct = gv is not None and gv.is_contiguous() and tuple(gv.shape) == ctx.pos_shapes[s]
            g = gv if direct else torch.zeros(ctx.pos_shapes[s], dtype=F32, device=dev)
            seeds, keep = ctx.drop
            if packs[s] is not None:
                # packed rows: dP[t] sums the rows (b, t) of the examples that have one (dmt_colsum_rows_packed; same dropout mask)
                L.call("dmt_colsum_rows_packed", ops.dt_code(dX[s].dtype), B, ctx.seq_T[s], d, ops.p(dX[s]), ops.p(packs[s].row_off),
                       ops.p(engine.seq_lens(batch, s)), 1.0, ops.p(g), int(seeds[s]), float(keep), 1 if ops.DETERMINISTIC else 0, ops.stream_ptr())
            elif 0.0 < keep < 1.0:
                L.call("dmt_colsum_drop", ops.dt_code(dX[s].dtype), B, ctx.seq_T[s] * d, ops.p(dX[s]), 1.0, ops.p(g), int(seeds[s]), float(keep),
                       1 if ops.DETERMINISTIC else 0, ops.stream_ptr())
            else:
                ops.colsum(dX[s].view(B, ctx.seq_T[s] * d), 1.0, out=g.view(-1)[: ctx.seq_T[s] * d])
            dpos.append(None if direct else g)
        engine.embedding_backward(batch, ctx.inv, ctx.seq_T, dX, dtar, dz, ctx.drop, packs)
        return (None, None, None, *dpos) + (None,) * (len(ctx.pos_leaves) - n_seq)      # (+ the anchor of a constant position table)


class AssembleFn(torch.autograd.Function):
    """z[:, off_s : off_s + d] = u_s  -- the tf.concat([features, interest_state]) of embedding_trans
    (mmoe_transformer_unbias.py:226-233), done in place on the gather output."""

    @staticmethod
    def forward(ctx, zbuf, off, d, *us):
        for s, u in enumerate(us):
            zbuf[:, off + s * d: off + (s + 1) * d].copy_(u)
        ctx.mark_dirty(zbuf)
        ctx.off, ctx.d, ctx.n = off, d, len(us)
        return zbuf

    @staticmethod
    def backward(ctx, dz):
        off, d = ctx.off, ctx.d
        return (dz, None, None, *[dz[:, off + s * d: off + (s + 1) * d] for s in range(ctx.n)])


def sp_units(spec):
    """(width of a task tower's input = the experts' last layer,)"""
    return (spec["hidden_units_bottom"][-1],)


class L2NormFn(torch.autograd.Function):
    """l2_norm of the reference (model/net/mmoe_transformer_unbias.py:42-60; added to the loss when wnd_wd > 1e-5, run_dnn.py:174-175).
    `anchor` is any leaf that requires grad (it only ties the node into the graph; its gradient is None)."""

    @staticmethod
    def forward(ctx, engine, batch, scale, anchor):
        st, sp = engine.store, engine.spec
        if st.shard is not None and st.shard[1] > 1:
            raise NotImplementedError("l2_norm on row-sharded tables")
        dev = st.device
        if engine._l2_mult is None:
            engine._l2_mult = torch.zeros(st.total_rows, dtype=torch.int32, device=dev)
        else:
            engine._l2_mult.zero_()
        out = torch.zeros(1, dtype=F32, device=dev)
        for (name, rows, dim, feat, _side) in sp["embedding_list"]:
            tf_name = "embedding_trans/%s/embedding" % name
            col = batch.feats[feat]
            base = st.table_rows[tf_name][0]
            seen = torch.zeros(rows // 32 + 1, dtype=torch.int32, device=dev)
            L.call("dmt_l2_unique_rows_count", batch.B, col.T, ops.p(col.idx), ops.p(col.lens), ops.p(st.table[tf_name]), rows, dim, ops.p(seen),
                   ops.p(out), ops.p(engine._l2_mult[base: base + rows]), ops.stream_ptr())
        ctx.engine, ctx.scale = engine, scale
        return out[0] * scale

    @staticmethod
    def backward(ctx, gout):
        ctx.engine._l2_coef = (gout.detach().to(F32) * ctx.scale).reshape(1).contiguous()
        return None, None, None, None


# ------------------------------------------------------------------------------------------------ engine
class DMTEngine:
    def __init__(self, spec: dict, store: VariableStore):
        self.spec, self.store = spec, store
        self.plan = _GatherPlan(spec, store)
        self._sparse = None          # (uniq_keys, n_uniq, grad_rows, cap) of the last backward
        self._l2_coef = None         # device scalar: a backward pass went through l2_norm (L2NormFn): its row gradients are still to be added
        self._l2_mult = None         # int32 [total rows]: embedding_list entries whose batch holds the row (L2NormFn.forward)
        self._ws = {}
        dev = store.device
        self.w_ctr = torch.tensor(spec["weight_ctr"], dtype=F32, device=dev)
        self.w_ecvr = torch.tensor(spec["weight_ecvr"], dtype=F32, device=dev)
        self.intermediates = {}
        self.dropout_step_seed = None    # int: dropout active with this per-step seed (is_train); None: off
        # fused self-attention block (dmt_mhsa_block_fwd: QKV projection + masked softmax attention + residual + LayerNorm in one
        # launch, T <= 64).  On by default since round 4; DMT_FUSED_MHSA=0 or Trainer(..., fused_mhsa=False) selects the
        # three-launch path (dmt_proj, dmt_attn_fwd, dmt_ln_fwd)
        self._use_mhsa = False
        self.use_mhsa = os.environ.get("DMT_FUSED_MHSA", "1") == "1"
        # the block's backward as ONE launch behind the LayerNorm gradient (dmt_mhsa_block_bwd) instead of dmt_attn_bwd + the dx GEMM.  Off by
        # default: measured 1.4x SLOWER than the two launches it replaces; DMT_FUSED_MHSA_BWD=1 selects it.
        self._use_mhsa_bwd = False
        self.use_mhsa_bwd = os.environ.get("DMT_FUSED_MHSA_BWD", "0") == "1"
        self.defer_sparse, self._pending_sparse = False, None     # GatherFn.backward leaves its work to finish_sparse_backward()
        self.seq_streams = os.environ.get("DMT_SEQ_STREAMS", "1") != "0"        # side streams for the behaviour sequences
        self.use_q1mem = os.environ.get("DMT_Q1MEM", "1") == "1"                # decoder attention over raw memory rows (dmt_q1mem.hip)
        self.use_heads_fused = os.environ.get("DMT_FUSED_HEADS", "1") == "1"    # towers + bias tower in one launch each way (dmt_heads.hip)
        self.use_mmoe_fused = os.environ.get("DMT_FUSED_MMOE", "1") == "1"      # fused expert-MLP + gate kernels (dmt_mmoe.hip)
        self.use_chain = True            # fused ff + ln kernels (dmt_chain2) where the geometry has one; False: GEMM + LN launches
        # packed rows (SeqPack): sequences whose padding is at least 10 % of their [B, T] grid are computed on their real rows only
        self.packed_rows = os.environ.get("DMT_PACKED_ROWS", "1") == "1"
        self.packed_rows_min_saving = 0.9
        self._last_packs = None
        self.step_state = ops.StepState()  # collected weight gradients / fork lane / long-row threshold of the step in flight (ops.StepState)
        self.kopts = ops.KernelOptions()  # attention / projection kernel choices of THIS engine (fp8 MFMA forward, long fused form, dmt_proj)

    @property
    def use_mhsa(self):
        return self._use_mhsa

    @use_mhsa.setter
    def use_mhsa(self, on):
        """The store rebuilds the block's weight images after every optimizer step only while the block is in use."""
        self._use_mhsa = bool(on)
        if self._use_mhsa and not self.store.mhsa_in_use:
            self.store.mhsa_in_use = True
            self.store.refresh_shadows()

    @property
    def use_mhsa_bwd(self):
        return self._use_mhsa_bwd

    @use_mhsa_bwd.setter
    def use_mhsa_bwd(self, on):
        """As use_mhsa: the store builds (and from then on rebuilds) the backward's weight images while an engine uses them -- here, on the
        caller's stream and before any sequence lane runs, not lazily from inside a lane (a lane would build all three images on ITS stream
        while the other lanes read theirs)."""
        self._use_mhsa_bwd = bool(on)
        if self._use_mhsa_bwd and self.store.mhsa_bwd and not self.store.mhsa_bwd_in_use:
            self.store.mhsa_bwd_in_use = True
            self.store.refresh_shadows()

    def gather_bytes(self, batch, seq_T) -> float:
        """Algorithmic HBM bytes of one gather launch: int32 indices read once, fp32 table rows for
        the valid ids on both paths, outputs written once in the compute dtype (padded slots excluded)."""
        esz = 4 if self.store.compute_dtype == F32 else 2
        B = batch.B
        total = 0.0
        for it in self.plan.items:
            col = batch.feats[it["feature"]]
            npos = B * col.T
            kinds = (1 if it["pooled_off"] >= 0 else 0) + (1 if it["seq_id"] >= 0 else 0)
            total += npos * 4 + npos * it["dim"] * 4 * kinds
            if it["pooled_off"] >= 0:
                total += B * it["dim"] * esz
        d = self.spec["d_model"]
        total += sum(B * t * d * esz for t in seq_T) + B * d * esz
        total += B * self.spec["feature_dimension"] * (4 + esz)
        return total

    # ---- small helpers
    def _w(self, name) -> Weight:
        return self.store.weight[name]

    def _lf(self, name):
        return self.store.leaf[name]

    @staticmethod
    def _wslice(w: Weight, a, b) -> Weight:
        return Weight(w.f32[:, a:b], w.lp[:, a:b] if w.lp is not None else None, w.lp_t[a:b, :] if w.lp_t is not None else None)

    def release(self):
        """Drop what the last pass left behind: `intermediates` holds graph-attached tensors whose autograd nodes point back at this engine
        (GatherFn keeps it in its ctx) -- a cycle through C++ graph edges that Python's collector cannot see, so an engine that is simply
        dropped keeps its last step's activations allocated (tens of GB at L = 200 in fp32: eighteen engines in one test module ran the
        device out of memory).  A long-lived engine never needs this; code that builds engines in a loop calls it (Trainer.close())."""
        self.intermediates = {}
        self._last_packs = None
        self._pending_sparse = None
        self._sparse = None
        self._l2_coef = None
        self._ws = {}

    @property
    def sparse(self):
        """(uniq_keys, n_uniq, grad_rows, cap) of the last backward.  The gradient of l2_norm (if the loss went through it) is added
        here, on first use: its backward node and the gather's run in no fixed order, both are done when anybody asks for the rows."""
        if self._l2_coef is not None and self._sparse is not None:
            coef, self._l2_coef = self._l2_coef, None
            uniq, n_uniq, rows, cap = self._sparse
            if int(cap) > 0:
                tm = self.store.fill_table_map(L.TableMap())
                L.call("dmt_l2_rows_add", C.byref(tm), ops.p(self.store.tab_p), ops.p(uniq), ops.p(n_uniq), int(cap), ops.p(self._l2_mult),
                       ops.p(coef), ops.p(rows), int(rows.shape[1]), ops.stream_ptr())
        return self._sparse

    @sparse.setter
    def sparse(self, value):
        self._sparse = value

    def l2_norm(self, batch: DeviceBatch, scale: float):
        """scale * sum over the embedding_list entries of l2_loss(E[distinct ids of the entry's feature]) (mmoe_transformer_unbias.py:42-60)
        with its gradient: scale * E[row] per entry on every distinct row, added to the sparse rows of the same backward pass."""
        return L2NormFn.apply(self, batch, float(scale), self._lf(trans_prefix(0) + "num_blocks_0/positionwise_feedforward/ln/gamma"))

    def position_tables(self):
        """The [maxlen_k, d_model] table added to every scaled sequence row (TransformerModel.py:60-69): the learned variable of each
        sequence (position_learn, dmt.conf:50) or the reference's sinusoid, a constant (position_sin_cos: TransformerModel_util.py:238-279,
        computed in float64 on the host as there)."""
        n = len(self.spec["attention_embed_pairs"])
        if self.spec.get("position_encoding_method", "position_learn") == "position_learn":
            return [self._lf(trans_prefix(i) + "positional_encoding_k_position_learn/embedding_position_learn") for i in range(n)]
        tab = getattr(self, "_sincos", None)
        if tab is None:
            E, maxlen = self.spec["d_model"], self.spec["maxlen_k"]
            i = np.arange(E)
            enc = np.arange(maxlen, dtype=np.float64)[:, None] / np.power(10000.0, (i - i % 2) / float(E))[None, :]
            enc[:, 0::2] = np.sin(enc[:, 0::2])
            enc[:, 1::2] = np.cos(enc[:, 1::2])
            tab = self._sincos = torch.tensor(enc.astype(np.float32), device=self.store.device)
        return [tab] * n

    # ---- stages
    def gather(self, batch: DeviceBatch):
        if self.spec.get("is_trans_input_by_mlp") and self.store.shard is not None:
            raise NotImplementedError("is_trans_input_by_mlp with row-sharded tables (the raw-row gather has no fetched-row form)")
        pos = self.position_tables()
        packs = [self.seq_pack(batch, s) for s in range(len(pos))]
        self._last_packs = packs
        self._l2_coef = None         # (a new forward pass: an l2 row term nobody asked the rows for belongs to the previous one)
        self._sparse = None          # ... and so do the previous backward's rows: the `sparse` getter must never add this pass's l2 term to them
        if not any(pl.requires_grad for pl in pos):
            # position_sin_cos: no input of the node requires a gradient, and autograd would not run its backward (the embedding gradient):
            # one more input, any leaf, ties it into the graph (its gradient is None)
            pos = pos + [self._lf(trans_prefix(0) + "num_blocks_0/positionwise_feedforward/ln/gamma")]
        outs = GatherFn.apply(self, batch, packs, *pos)
        pos = pos[: len(packs)]
        n = len(pos)
        return list(outs[:n]), outs[n], outs[n + 1]

    # ---- packed rows
    def seq_lens(self, batch, s):
        """Device lengths of sequence s: those of its LAST field pair (mask / lens of mmoe_transformer.py:137-142)."""
        return batch.feats[self.spec["attention_embed_pairs"][s][-1][0]].lens

    def seq_pack(self, batch, s):
        """The SeqPack of sequence s of this batch, or None when the sequence is computed in the dense [B, T, d] layout.  Packed rows
        are taken when every kernel of the sequence's path has its packed form (bf16, the fused self-attention block, the raw-memory
        decoder attention, the fused FFN), the lengths are known on the host and within [1, T], and the padding is worth skipping
        (R <= packed_rows_min_saving * B * T).  Training and inference alike: the layout is a property of the forward graph."""
        key = (id(self), s, self.packed_rows, self.use_mhsa, self.use_q1mem, self.use_chain)       # (a batch may be shared by engines / switch settings)
        if key in batch._packs:
            return batch._packs[key]
        pack = None
        spec, store = self.spec, self.store
        d, H = spec["d_model"], spec["num_heads"]
        pairs = spec["attention_embed_pairs"][s]
        col = batch.feats[pairs[-1][0]]
        T = max(batch.feats[uf].T for (uf, _i) in pairs)
        blk = trans_prefix(s) + "num_blocks_0/"
        ok = (self.packed_rows and store.compute_dtype == torch.bfloat16 and store.device.type == "cuda" and store.shard is None
              and self.use_mhsa and store.mhsa.get(blk + "self-attention/") is not None and ops.mhsa_supported(d, H, T, batch.B)
              and self.use_q1mem and store.q1mem.get(blk + "vanilla_attention/") is not None and ops.q1mem_supported(d, H, T)
              and self.use_chain and store.chain.get(blk + "positionwise_feedforward/") is not None
              and T >= 8 and SeqPack.eligible(col.lens_host, T) and not spec.get("is_trans_input_by_mlp"))
        if ok:
            R = int(col.lens_host.astype(np.int64).sum())
            if R <= self.packed_rows_min_saving * batch.B * T:
                pack = SeqPack(col.lens_host, T, store.device)
        batch._packs[key] = pack
        return pack

    def gather_raw(self, batch: DeviceBatch):
        """generate_data's outputs: unscaled seq_emb / tar_emb ([0;E] lookups, no positions); no gradient path to the
        tables (API-parity helper; the training path uses gather())."""
        self._replicated_only("gather_raw")
        spec, plan, store = self.spec, self.plan, self.store
        dev, cdt = store.device, store.compute_dtype
        B, d = batch.B, spec["d_model"]
        n_seq = len(spec["attention_embed_pairs"])
        seq_T = [max(batch.feats[uf].T for (uf, _i) in pairs) for pairs in spec["attention_embed_pairs"]]
        X = [torch.empty((B, seq_T[s], d), dtype=cdt, device=dev) for s in range(n_seq)]
        tar = torch.empty((B, d), dtype=cdt, device=dev)
        desc = L.GatherDesc()
        items = [it for it in plan.items if it["seq_id"] >= 0]
        desc.B, desc.n_features = B, len(items)
        for i, it in enumerate(items):
            col = batch.feats[it["feature"]]
            f = desc.feat[i]
            f.table = store.table[it["table"]].data_ptr()
            f.rows, f.dim = it["rows"], it["dim"]
            f.idx, f.wts, f.lens, f.T = col.idx.data_ptr(), None, col.lens.data_ptr(), col.T
            f.pooled_off, f.seq_id, f.seq_off, f.group, f.inv_wsum = -1, it["seq_id"], it["seq_off"], it["group"], None
        desc.n_seq = n_seq
        for s in range(n_seq):
            desc.seq_out[s], desc.seq_T[s], desc.pos[s] = X[s].data_ptr(), seq_T[s], None
        desc.tar_out, desc.d_model, desc.seq_scale = tar.data_ptr(), d, 1.0
        desc.pooled, desc.ld_pooled, desc.dense, desc.n_dense = None, 0, None, 0
        desc.out_dtype = ops.dt_code(cdt)
        L.call("dmt_gather_fwd", C.byref(desc), ops.stream_ptr())
        return X, tar

    def gather_pooled(self, batch: DeviceBatch):
        """Inference-only gather WITHOUT the sequence rows: target embedding tar [B, d] and the MMoE input buffer zbuf (dense
        features, pooled id embeddings, bias-tower embeddings).  Used by the serving path, where the behaviour sequences of the
        one user are encoded once (serving.CandidateScorer) and only these per-candidate pieces are needed for every row."""
        self._replicated_only("gather_pooled")
        spec, plan, store = self.spec, self.plan, self.store
        dev, cdt = store.device, store.compute_dtype
        B, d = batch.B, spec["d_model"]
        tar = torch.empty((B, d), dtype=cdt, device=dev)
        zbuf = torch.zeros((B, plan.ldz), dtype=cdt, device=dev)
        inv = torch.empty((len(plan.items), B), dtype=F32, device=dev)
        desc = L.GatherDesc()
        desc.B, desc.n_features = B, len(plan.items)
        for i, it in enumerate(plan.items):
            col = batch.feats[it["feature"]]
            f = desc.feat[i]
            f.table = store.table[it["table"]].data_ptr()
            f.rows, f.dim = it["rows"], it["dim"]
            f.idx = col.idx.data_ptr()
            f.wts = col.wts.data_ptr() if col.wts is not None else None
            f.lens, f.T = col.lens.data_ptr(), col.T
            is_target = it["seq_id"] == L.DMT_SEQ_TARGET
            f.pooled_off, f.group = it["pooled_off"], it["group"]
            f.seq_id, f.seq_off = (it["seq_id"], it["seq_off"]) if is_target else (-1, 0)
            f.inv_wsum = inv[i].data_ptr()
        desc.n_seq = 0
        desc.tar_out, desc.d_model, desc.seq_scale = tar.data_ptr(), d, (1.0 if spec.get("is_trans_input_by_mlp") else float(d) ** 0.5)
        desc.pooled, desc.ld_pooled = zbuf.data_ptr(), plan.ldz
        desc.dense, desc.n_dense = batch.dense.data_ptr(), spec["feature_dimension"]
        desc.out_dtype = ops.dt_code(cdt)
        desc.seq_drop_keep = 1.0
        L.call("dmt_gather_fwd", C.byref(desc), ops.stream_ptr())
        return tar, zbuf

    def _attn_drop(self, stream):
        rate = self.spec.get("dropout_rate", 0.0)
        if self.dropout_step_seed is None or not rate:
            return 0, 1.0
        return ops.site_seed(self.dropout_step_seed, stream), 1.0 - rate

    def mha_self(self, x, lens, blk, stream=2, pack=None):
        """multihead_attention(x, x, x, lens, lens) (TransformerModel_util.py:160-209), x: [B,T,d] (pack: [1, R, d] packed rows)."""
        d, H = self.spec["d_model"], self.spec["num_heads"]
        a = blk + "self-attention/"
        img = self.store.mhsa.get(a) if x.dtype == torch.bfloat16 else None
        if pack is not None or (img is not None and self.use_mhsa and ops.mhsa_supported(d, H, x.shape[1], x.shape[0])):
            seed, keep = self._attn_drop(stream)
            img_b = None
            if self.use_mhsa_bwd and img is not None and self.store.mhsa_bwd_in_use:
                img_b = self.store.mhsa_bwd.get(a)
            return ops.MhsaBlockFn.apply(x, self._lf(a + "qkv_kernel"), self._lf(a + "qkv_bias"), self._w(a + "qkv_kernel"),
                                         self._lf(a + "ln/gamma"), self._lf(a + "ln/beta"), lens, H, img, seed, keep, 1e-8, pack, img_b)
        s1 = ops.SelfAttnBlockFn.apply(x, self._lf(a + "qkv_kernel"), self._lf(a + "qkv_bias"), self._w(a + "qkv_kernel"), lens, H,
                                       *self._attn_drop(stream), self.kopts)
        return ops.layer_norm(s1, self._lf(a + "ln/gamma"), self._lf(a + "ln/beta"))

    def mha_cross(self, q_in, mem, q_lens, k_lens, blk, stream=3, pack=None):
        """multihead_attention(q, mem, mem, q_lens, k_lens) with scope 'vanilla_attention' (pack: mem is [1, R, d] packed rows)."""
        d, H = self.spec["d_model"], self.spec["num_heads"]
        a = blk + "vanilla_attention/"
        wl, bl, w = self._lf(a + "qkv_kernel"), self._lf(a + "qkv_bias"), self._w(a + "qkv_kernel")
        q = ops.linear(q_in, wl[:, :d], bl[:d], self._wslice(w, 0, d))
        wv_aug = self.store.q1mem.get(a) if (self.use_q1mem and q_in.dtype == torch.bfloat16) else None
        if pack is not None or (wv_aug is not None and q_lens is None and q_in.shape[1] == 1 and ops.q1mem_supported(d, H, mem.shape[1]) and
                                mem.stride(2) == 1 and mem.stride(1) % 8 == 0 and mem.stride(0) % 8 == 0 and mem.data_ptr() % 16 == 0):
            # one query per example: attend over the raw memory rows, no K / V projection of the memory (dmt_q1mem.hip)
            seed, keep = self._attn_drop(stream)
            s = ops.CrossQ1Fn.apply(q.reshape(-1, d), mem, q_in.reshape(-1, d), k_lens, w, wl, bl, wv_aug, H, seed, keep, pack).unsqueeze(1)
            return ops.layer_norm(s, self._lf(a + "ln/gamma"), self._lf(a + "ln/beta"))
        kv = ops.linear(mem, wl[:, d:], bl[d:], self._wslice(w, d, 3 * d))
        s = ops.AttnFn.apply(q, kv, q_in, q_lens, k_lens, H, d, False, *self._attn_drop(stream), self.kopts)
        return ops.layer_norm(s, self._lf(a + "ln/gamma"), self._lf(a + "ln/beta"))

    def ff(self, x, ffs):
        """ff(inputs, [d_ff, d_model]) (TransformerModel_util.py:212-235)."""
        chain = self.store.chain.get(ffs) if x.dtype == torch.bfloat16 else None
        if chain is not None and self.use_chain:
            return ops.FFNLNChainFn.apply(x, self._lf(ffs + "dense/kernel"), self._lf(ffs + "dense/bias"), self._lf(ffs + "dense_1/kernel"),
                                          self._lf(ffs + "dense_1/bias"), self._lf(ffs + "ln/gamma"), self._lf(ffs + "ln/beta"), chain, 1e-8)
        s = ops.FFNFn.apply(x, self._lf(ffs + "dense/kernel"), self._lf(ffs + "dense/bias"), self._lf(ffs + "dense_1/kernel"),
                            self._lf(ffs + "dense_1/bias"), self._w(ffs + "dense/kernel"), self._w(ffs + "dense_1/kernel"))
        return ops.layer_norm(s, self._lf(ffs + "ln/gamma"), self._lf(ffs + "ln/beta"))

    def encode_prepared(self, x, lens, i, pack=None):
        """TransformerModel.encode after the input prep (x = sqrt(d)*seq_emb + P, fused into the gather)."""
        # TransformerModel.py:101 dropout(enc): already applied by the gather (GatherFn), stream id 10 * i + 0
        for j in range(int(self.spec.get("num_blocks_encode", 1))):          # TransformerModel.py:104-121 (dmt.conf: one block)
            blk = trans_prefix(i) + "num_blocks_%d/" % j
            x = self.mha_self(x, lens, blk, 10 * i + 2 + 1000 * j, pack=pack)        # (an independent attention-dropout mask per block)
            x = self.ff(x, blk + "positionwise_feedforward/")
        return x

    def decode_prepared(self, y, mem, lens, i, pack=None):
        """TransformerModel.decode after the input prep (y = sqrt(d)*tar[:,None,:])."""
        y = ops.dropout(y, self.spec.get("dropout_rate", 0.0), self.dropout_step_seed, 10 * i + 1)     # TransformerModel.py:151
        ffs = "positionwise_feedforward/" if self.spec.get("tie_ffn", True) else "positionwise_feedforward_dec/"
        for j in range(int(self.spec.get("num_blocks_decode", 1))):          # TransformerModel.py:154-169 (dmt.conf: one block)
            blk = trans_prefix(i) + "num_blocks_%d/" % j
            y = self.mha_cross(y, mem, None, lens, blk, 10 * i + 3 + 1000 * j, pack=pack)
            y = self.ff(y, blk + ffs)
        return y

    def decode_shared(self, y, mem1, k_lens, i):
        """decode_prepared for B queries against ONE shared memory mem1 [1, T, d] (serving: every candidate row of a request has
        the same user): K | V are projected once and broadcast through a zero batch stride.  Inference only (no dropout)."""
        d, H = self.spec["d_model"], self.spec["num_heads"]
        ffs = "positionwise_feedforward/" if self.spec.get("tie_ffn", True) else "positionwise_feedforward_dec/"
        for j in range(int(self.spec.get("num_blocks_decode", 1))):
            blk = trans_prefix(i) + "num_blocks_%d/" % j
            a = blk + "vanilla_attention/"
            wl, bl, w = self._lf(a + "qkv_kernel"), self._lf(a + "qkv_bias"), self._w(a + "qkv_kernel")
            q = ops.linear(y, wl[:, :d], bl[:d], self._wslice(w, 0, d))
            kv = ops.linear(mem1, wl[:, d:], bl[d:], self._wslice(w, d, 3 * d)).expand(y.shape[0], -1, -1)
            s = ops.AttnFn.apply(q, kv, y, None, k_lens, H, d, False, 0, 1.0, self.kopts)
            s = ops.layer_norm(s, self._lf(a + "ln/gamma"), self._lf(a + "ln/beta"))
            y = self.ff(s, blk + ffs)
        return y

    def interest_blocks(self, us, tars_scaled):
        """The d_model-wide blocks of interest_state: user_stat per sequence, each followed by the RAW target embedding when
        is_trans_out_concat_item (mmoe_transformer_unbias.py:212-219; the gather hands out the decoder's copy, scaled by sqrt(d_model):
        undone here with one rounding in the compute dtype)."""
        if not self.spec.get("is_trans_out_concat_item"):
            return us
        if not isinstance(tars_scaled, (list, tuple)):
            tars_scaled = [tars_scaled] * len(us)
        inv = 1.0 / float(self.spec["d_model"]) ** 0.5
        raws, seen = [], {}
        for t in tars_scaled:                      # (one division per distinct tensor: without the input MLP all sequences share it)
            if id(t) not in seen:
                seen[id(t)] = t * inv
            raws.append(seen[id(t)])
        if not self.spec.get("is_trans_out_by_mlp"):
            return [t for u, raw in zip(us, raws) for t in (u, raw)]
        out = []
        for i, (u, raw) in enumerate(zip(us, raws)):      # :216-217: tf.layers.dense([user_stat, tar_sku_emb], d_model, name='dense_trans_concat_' + stag)
            tp = "embedding_trans/trans_sequence_%d/dense_trans_concat_sequence_%d/" % (i, i)
            out.append(ops.linear(torch.cat([u, raw], -1), self._lf(tp + "kernel"), self._lf(tp + "bias"), self._w(tp + "kernel")))
        return out

    def input_mlp(self, i, x_raw, tar_raw):
        """is_trans_input_by_mlp (mmoe_transformer_unbias.py:196-198): the RAW rows of sequence i and the raw target rows through
        'dense_trans_seq_<stag>' / 'dense_trans_sku_<stag>', then the prep the gather skipped for this variant (TransformerModel.py:96-101:
        * sqrt(d_model), + positions, dropout of the block input; the decoder's query is scaled the same way) -> (x, scaled target)."""
        scale = float(self.spec["d_model"]) ** 0.5
        tp = "embedding_trans/trans_sequence_%d/" % i
        ks, kt = tp + "dense_trans_seq_sequence_%d/" % i, tp + "dense_trans_sku_sequence_%d/" % i
        x = ops.linear(x_raw, self._lf(ks + "kernel"), self._lf(ks + "bias"), self._w(ks + "kernel"))
        x = ops.ScaleAddPosFn.apply(x, self.position_tables()[i], scale)
        x = ops.dropout(x, self.spec.get("dropout_rate", 0.0), self.dropout_step_seed, 10 * i + 0)
        tar = ops.linear(tar_raw, self._lf(kt + "kernel"), self._lf(kt + "bias"), self._w(kt + "kernel")) * scale
        return x, tar

    def decoder_query(self, tar):
        """The decoder's one-step query from the scaled target rows: + row 0 of the sinusoid (sin(0) on the even columns, cos(0) on the
        odd ones) when is_decoder_add_pos_emb (TransformerModel.py:148-149: dec += positional_encoding(dec, maxlen_q); dmt.conf: false)."""
        if not self.spec.get("is_decoder_add_pos_emb"):
            return tar
        pe0 = getattr(self, "_pe0", None)
        if pe0 is None or pe0.dtype != tar.dtype:
            pe0 = self._pe0 = (torch.arange(self.spec["d_model"], device=tar.device) % 2).to(tar.dtype)
        return tar + pe0

    def embedding_trans(self, batch: DeviceBatch):
        X, tar, zbuf = self.gather(batch)
        packs = self._last_packs
        n_seq = len(self.spec["attention_embed_pairs"])
        tars_scaled = [tar] * n_seq
        # The behaviour sequences are independent between the gather and the assembly of z: with seq_streams each runs on its own
        # stream (autograd replays the backward of every op on the stream of its forward), so the launch-latency-bound B-row kernels
        # of one sequence's decoder fill the tails of another's big kernels.
        main = ops.cur_stream(self.store.device) if (self.seq_streams and X[0].is_cuda) else None
        pairs_all = self.spec["attention_embed_pairs"]
        us, order = [None] * n_seq, list(range(n_seq))
        tars = list(FanOutFn.apply(tar, n_seq) if (tar.requires_grad and n_seq > 1) else (tar,) * n_seq)
        if self.spec.get("is_trans_input_by_mlp"):
            for i in range(n_seq):
                X[i], tars[i] = self.input_mlp(i, X[i], tars[i])
            tars_scaled = list(tars)
        tars = [self.decoder_query(t) for t in tars]
        if main is not None:
            # Sequence 0 stays on the compute stream; the others start from ONE event (the gathered inputs), not behind sequence 0.
            # (Measured: issue order -- by length, either way -- moves the step by < 1 %; putting the compute stream's sequence
            # anywhere but first costs 8 %.  The big kernels fill the chip on their own: what overlaps is the small-kernel stretches.)
            side = self._seq_stream_pool(n_seq)
            ready = torch.cuda.Event()
            ready.record(main)
        for pos, i in enumerate(order):
            lens = batch.feats[pairs_all[i][-1][0]].lens          # mask / lens come from the LAST pair (mmoe_transformer.py:137-142)
            st = side[pos] if main is not None else None
            if st is not None:
                st.wait_event(ready)
                X[i].record_stream(st)            # (allocated on the compute stream, read on this one)
                tars[i].record_stream(st)
                if packs[i] is not None:
                    packs[i].buf.record_stream(st)
                with torch.cuda.stream(st):
                    mem = self.encode_prepared(X[i], lens, i, pack=packs[i])
                    y = self.decode_prepared(tars[i].unsqueeze(1), mem, lens, i, pack=packs[i])
                    y.record_stream(main)
                    mem.record_stream(main)
            else:
                mem = self.encode_prepared(X[i], lens, i, pack=packs[i])
                y = self.decode_prepared(tars[i].unsqueeze(1), mem, lens, i, pack=packs[i])
            us[i] = y.squeeze(1)
            self.intermediates["memory_%d" % i] = mem          # ([1, R, d] when the sequence ran on packed rows: pack_%d)
            self.intermediates["pack_%d" % i] = packs[i]
        if main is not None:
            for st in side:
                if st is not None:
                    main.wait_stream(st)
            # the junction: from here to the first long backward kernel the compute stream runs one B-row kernel after the other
            # (MMoE layer 0, experts, towers, loss and back) and most of the chip is idle -- Trainer._catch_up_early starts there
            self.junction_event = torch.cuda.Event()
            self.junction_event.record(main)
        z = AssembleFn.apply(zbuf, self.plan.interest_off, self.spec["d_model"], *self.interest_blocks(us, tars_scaled))
        self.intermediates["zbuf"] = z
        return z

    def _seq_stream_pool(self, n):
        if getattr(self, "_seq_streams", None) is None or len(self._seq_streams) != n:
            from . import streams
            pool = list(streams.lanes(self.store.device)["seq"])          # queue-bound lanes (streams.py); more sequences: plain streams
            while len(pool) < n - 1:
                pool.append(torch.cuda.Stream(self.store.device))
            self._seq_streams = [None] + pool[: n - 1]
        return self._seq_streams

    def expert_gate(self, z, want_mix=False, z_is_engine_buffer=False):
        """Per-task mixtures; want_mix: as ONE [T, B, U] tensor (for heads()), else a list of [B, U] (reference shape).
        z_is_engine_buffer: z is (a view of) this engine's own zero-initialised zbuf, whose columns past K hold finite values (the bias
        tower's inputs) -- only then may the layer-0 GEMM read the pad column K .. ceil8(K) against its zero weight row.  A caller's own
        tensor (the reference-named facade) may hold NaN / uninitialised memory there: 0 * NaN would poison every output."""
        sp = self.spec
        E, T, units = sp["num_experts"], sp["num_tasks"], sp["hidden_units_bottom"]
        K = self.plan.K
        zin = z if z.shape[1] == K else z[:, :K]      # inference() hands over the already split [B, K] view
        fused = self.use_mmoe_fused and ops.mmoe_experts_supported(units, E, T, zin.dtype)
        # (fused expert kernels: their backward hands d g1 over already times the relu gradient of this layer -- no pass of its own)
        g1 = ops.linear(zin, self._lf("mmoe_layers/l0_cat_weights"), self._lf("mmoe_layers/l0_cat_biases"),
                        self._w("mmoe_layers/l0_cat_weights"), act_ncols=E * units[0], x_pad_finite=bool(z_is_engine_buffer), relu_grad_by_consumer=fused)
        # [B, E * u0]: the four experts' layer-0 outputs side by side | both gates' logits
        if fused:
            # fused expert-MLP + gate kernels: layers 1-2 of every expert, the gate softmaxes and the mixtures in one launch
            names = [["mmoe_layers/expert-%d/expert-layer-%d/" % (e, li) for e in range(E)] for li in (1, 2)]
            mix, gates = ops.MmoeExpertsFn.apply(g1, [self._w(n + "weights") for n in names[0]], [self._w(n + "weights") for n in names[1]],
                                                 [self._lf(n + "weights") for n in names[0]], [self._lf(n + "biases") for n in names[0]],
                                                 [self._lf(n + "weights") for n in names[1]], [self._lf(n + "biases") for n in names[1]], E, T, True)
            self.intermediates["gates"] = gates
            return mix if want_mix else list(ops.Unbind0Fn.apply(mix))
        expert, glogit = ops.split_cols(g1, 0, E * units[0], E * units[0], E * units[0] + T * E)
        for li in range(1, len(units)):
            nms = ["mmoe_layers/expert-%d/expert-layer-%d/" % (e, li) for e in range(E)]
            expert = ops.expert_layer(expert, [self._w(n + "weights") for n in nms], [self._lf(n + "weights") for n in nms],
                                      [self._lf(n + "biases") for n in nms])
        mix, gates = ops.MixFn.apply(expert, glogit, E, units[-1], T)
        self.intermediates["gates"] = gates
        return mix if want_mix else list(ops.Unbind0Fn.apply(mix))

    def heads(self, mix, z_bias):
        """Task towers + position-bias tower in one launch (dmt_heads_fwd): ((click, order), y_bias), each [B, 1] fp32."""
        sp = self.spec
        towers = []
        for name in ("click", "order")[: sp["num_tasks"]]:
            f, o = "%s/%s-fc-0/" % (name, name), "%s/%s-output/" % (name, name)
            towers.append((self._w(f + "weights"), self._lf(f + "weights"), self._lf(f + "biases"),
                           self._w(o + "weights"), self._lf(o + "weights"), self._lf(o + "biases")))
        bias = [(self._w("layer_bias%d/kernel" % l), self._lf("layer_bias%d/kernel" % l), self._lf("layer_bias%d/bias" % l)) for l in range(3)]
        rates = sp.get("dropout_rate_bias", [0.0, 0.0])
        drops = []
        for l in range(2):
            on = self.dropout_step_seed is not None and rates[l]
            drops.append((ops.site_seed(self.dropout_step_seed, 100 + l), 1.0 - rates[l]) if on else (0, 1.0))
        outs = ops.HeadsFn.apply(mix, z_bias, towers, bias, tuple(drops))
        return tuple(outs[:-1]), outs[-1]

    def build_tower(self, x, name):
        sp = self.spec
        h = x
        for li in range(len(sp["hidden_units_task"])):
            nm = "%s/%s-fc-%d/" % (name, name, li)
            h = ops.linear(h, self._lf(nm + "weights"), self._lf(nm + "biases"), self._w(nm + "weights"), relu=True)
        nm = "%s/%s-output/" % (name, name)
        return ops.linear(h, self._lf(nm + "weights"), self._lf(nm + "biases"), self._w(nm + "weights"), out_dtype=F32)

    def embedding_mlp_bias(self, z):
        sp = self.spec
        h = z if z.shape[1] == self.plan.bias_width else z[:, self.plan.bias_off: self.plan.bias_off + self.plan.bias_width]
        n = len(sp["hidden_units_bias"])
        for li in range(n):
            nm = "layer_bias%d/" % li
            h = ops.linear(h, self._lf(nm + "kernel"), self._lf(nm + "bias"), self._w(nm + "kernel"), relu=True)
            h = ops.dropout(h, sp.get("dropout_rate_bias", [0.0] * n)[li], self.dropout_step_seed, 100 + li)   # :274-278
        nm = "layer_bias%d/" % n
        return ops.linear(h, self._lf(nm + "kernel"), self._lf(nm + "bias"), self._w(nm + "kernel"), out_dtype=F32)

    def inference(self, batch: DeviceBatch, is_predict=False):
        z = self.embedding_trans(batch)
        if is_predict:
            tasks = self.expert_gate(z, z_is_engine_buffer=True)
            return tuple(self.build_tower(m, nm) for m, nm in zip(tasks, ("click", "order")))
        # MMoE input | bias-tower input: one autograd node for both column slices of z
        plan = self.plan
        if plan.bias_width == plan.K:      # (degenerate layout: keep the two views distinguishable by the plain path)
            z_main, z_bias = z[:, :plan.K], z[:, plan.bias_off: plan.bias_off + plan.bias_width]
        else:
            z_main, z_bias = ops.split_cols(z, 0, plan.K, plan.bias_off, plan.bias_off + plan.bias_width)
        if self.use_heads_fused and ops.heads_supported(sp_units(self.spec)[0], self.spec["hidden_units_task"], plan.bias_width,
                                                        self.spec["hidden_units_bias"], self.spec["num_tasks"], z.dtype):
            mix = self.expert_gate(z_main, want_mix=True, z_is_engine_buffer=True)
            return self.heads(mix, z_bias)
        tasks = self.expert_gate(z_main, z_is_engine_buffer=True)
        logits = tuple(self.build_tower(m, nm) for m, nm in zip(tasks, ("click", "order")))
        return logits, self.embedding_mlp_bias(z_bias)

    def loss_unbias(self, out, mask, method=None, ctr_rel=None):
        (c, o), yb = out
        sp = self.spec
        method = sp["loss_unbias_method"] if method is None else method
        ctr_rel = sp["loss_ctr_rel_method"] if ctr_rel is None else ctr_rel
        loss, pc, pv = ops.LossUnbiasFn.apply(c, o, yb, mask, self.w_ctr, self.w_ecvr, sp["loss_weight"],
                                              1 if method == "two_head_multiply" else 0, 1 if ctr_rel == "ctr_rel" else 0)
        return loss, pc, pv

    # ---- sparse embedding gradient
    def _buf(self, key, shape, dtype):
        t = self._ws.get(key)
        n = int(np.prod(shape))
        if t is None or t.numel() < n or t.dtype != dtype:
            t = torch.empty(n, dtype=dtype, device=self.store.device)
            self._ws[key] = t
        return t[:n].view(shape)

    def _embgrad_desc(self, batch):
        """Index-only part of the embedding-gradient descriptor (gradient pointers are filled in by backward)."""
        plan, store, spec = self.plan, self.store, self.spec
        desc = L.EmbGradDesc()
        desc.B, desc.n_features = batch.B, len(plan.items)
        ebase = 0
        for i, it in enumerate(plan.items):
            col = batch.feats[it["feature"]]
            f = desc.feat[i]
            f.rows, f.dim = it["rows"], it["dim"]
            f.idx = col.idx.data_ptr()
            f.wts = col.wts.data_ptr() if col.wts is not None else None
            f.lens = col.lens.data_ptr()
            f.T = col.T
            f.pooled_off, f.seq_id, f.seq_off, f.group = it["pooled_off"], it["seq_id"], it["seq_off"], it["group"]
            desc.row_base[i] = it["row_base"]
            desc.entry_base[i] = ebase
            kinds = (1 if it["pooled_off"] >= 0 else 0) + (1 if it["seq_id"] >= 0 else 0)
            ebase += kinds * batch.B * col.T
        desc.entry_base[len(plan.items)] = ebase
        desc.total_rows = store.total_rows
        desc.d_model = spec["d_model"]
        desc.seq_scale = 1.0 if spec.get("is_trans_input_by_mlp") else float(spec["d_model"]) ** 0.5
        return desc, ebase

    def prepare(self, batch):
        """Index-only preprocessing of a batch, done BEFORE the forward pass: the (table row, entry) pairs of every
        id are stably sorted by row and segmented.  The distinct rows are (a) what the exact lazy Adam must catch up
        before the gather reads them and (b) the segments of the backward reduction."""
        prep = getattr(batch, "_prep", None)
        if prep is not None:
            return prep
        desc, n = self._embgrad_desc(batch)
        st = ops.stream_ptr()
        dev = self.store.device
        keys = self._buf("keys", (n,), torch.int32)
        keys_s = torch.empty((n,), dtype=torch.int32, device=dev)
        vals_s = torch.empty((n,), dtype=torch.int32, device=dev)
        L.call("dmt_embgrad_keys", C.byref(desc), ops.p(keys), None, st)             # (values = entry numbers: the sort writes them itself)
        uniq, n_uniq, seg = self.sort_segments(keys, None, keys_s, vals_s, n, own=True)      # (kept by the batch: no copies)
        cap = min(n, self.store.total_rows)
        prep = dict(desc=desc, n=n, keys_s=keys_s, vals_s=vals_s, seg=seg, uniq=uniq[:cap], n_uniq=n_uniq, cap=cap)
        batch._prep = prep
        return prep

    # ---- row-sharded tables: fetch the rows a batch reads from their owners (BASELINE configs[3])
    def fetch_rows(self, batch, opt=None, plan=None):
        """Index exchange + row return.  The batch's distinct global rows (engine.prepare) go to their owners (row % W) by
        all_to_all; every owner first replays the pending lazy-Adam updates of the requested rows (opt.catch_up), gathers them
        (dmt_rows_gather) and sends them back; the entries' ids are rewritten into slots of the returned row cache
        (dmt_entry_slots).  The cache lives in batch._prep["cache"] until the next fetch."""
        from . import parallel
        prep = self.prepare(batch)
        store = self.store
        uniq = prep["uniq"]
        if plan is not None:         # Trainer.plan_exchange already moved the ids (the same lists serve the gradient rows' way back)
            n, perm, recv_k, send_splits, recv_splits = plan["n"], plan["perm"], plan["recv_k"], plan["send_splits"], plan["recv_splits"]
        else:
            n = int(prep["n_uniq"].item())
            perm, recv_k, send_splits, recv_splits = parallel.request_rows(uniq, n)
        R = recv_k.numel()
        if opt is not None and opt.global_step > 0 and R > 0:
            # replay on DISTINCT rows only: a hot row is requested by several ranks, and two wavefronts replaying the same row at once
            # tear each other's p / m / v (seen as a 1e-5 loss difference in one run out of three at two ranks)
            if plan is not None and plan.get("keys_s") is not None:
                opt.catch_up(plan["uniq2"], plan["n_uniq2"], R)
            else:
                ku = torch.unique(recv_k)
                opt.catch_up(ku.to(recv_k.dtype), torch.tensor([ku.numel()], dtype=torch.int32, device=recv_k.device), int(ku.numel()))
        tm = store.fill_table_map(L.TableMap())
        D = self.plan.max_dim
        rows_out = torch.empty((max(R, 1), D), dtype=F32, device=store.device)
        L.call("dmt_rows_gather", C.byref(tm), ops.p(store.tab_p), ops.p(recv_k), R, ops.p(rows_out), D, ops.stream_ptr())
        cache = parallel.return_rows(rows_out[:R], n, send_splits, recv_splits)
        if cache.shape[0] == 0:
            cache = torch.zeros((1, D), dtype=F32, device=store.device)
        # slot of distinct row u = its position in the owner-grouped order
        pos = torch.empty((max(n, 1),), dtype=torch.int32, device=store.device)
        pos[perm] = torch.arange(n, dtype=torch.int32, device=store.device)
        slots = torch.empty((prep["n"],), dtype=torch.int32, device=store.device)
        L.call("dmt_entry_slots", C.byref(prep["desc"]), ops.p(prep["keys_s"]), ops.p(prep["vals_s"]), ops.p(prep["seg"]), ops.p(pos), prep["n"],
               ops.p(slots), ops.stream_ptr())
        ebase = [int(prep["desc"].entry_base[i]) for i in range(len(self.plan.items))]
        prep["cache"] = (cache.contiguous(), slots, ebase)
        return prep["cache"]

    def _replicated_only(self, what):
        if self.store.shard is not None and self.store.shard[1] > 1:
            raise RuntimeError("%s reads whole embedding tables: not available with row-sharded tables (serve from a replicated store)" % what)

    def row_cache(self, batch):
        prep = getattr(batch, "_prep", None)
        if prep is None or "cache" not in prep:
            raise RuntimeError("row-sharded tables: call Trainer.sync_rows(batch) (engine.fetch_rows) before the forward pass")
        return prep["cache"]

    def _input_dropout(self, n_seq):
        """(per-sequence site seeds, keep probability) of the block-input dropout; keep = 0.0 when dropout is off."""
        rate = self.spec.get("dropout_rate", 0.0)
        if self.dropout_step_seed is None or not rate:
            return [0] * n_seq, 0.0
        return [ops.site_seed(self.dropout_step_seed, 10 * s + 0) for s in range(n_seq)], 1.0 - rate

    def finish_sparse_backward(self):
        """The deferred tail of backward (GatherFn.backward with defer_sparse): position-table gradients, zero + segment-reduce of the
        embedding-gradient rows -> self.sparse.  Runs on the CURRENT stream; the caller orders it behind backward."""
        pend, self._pending_sparse = self._pending_sparse, None
        if pend is None:
            return False
        ctx, grads = pend
        cur = ops.cur_stream(self.store.device)
        for g in grads:
            if g is not None and g.is_cuda:
                g.record_stream(cur)
        for pk in ctx.packs:
            if pk is not None:
                pk.buf.record_stream(cur)
        out = GatherFn._finish(ctx, grads)
        assert all(o is None for o in out)        # (position gradients went straight into the arena)
        return True

    def embedding_backward(self, batch, inv, seq_T, dX, dtar, dz, drop=None, packs=None):
        plan = self.plan
        prep = self.prepare(batch)
        desc, n = prep["desc"], prep["n"]
        for s in range(len(dX)):                 # (the descriptor is cached on the batch: set both ways every time)
            pk = packs[s] if packs is not None else None
            desc.seq_row_off[s] = pk.row_off.data_ptr() if pk is not None else None
            desc.seq_row_len[s] = self.seq_lens(batch, s).data_ptr() if pk is not None else None
        seeds, keep = drop if drop is not None else ([0] * len(dX), 0.0)
        for s in range(len(dX)):
            desc.seq_drop_seed[s] = seeds[s]
        desc.seq_drop_keep = keep
        for i in range(len(plan.items)):
            desc.feat[i].inv_wsum = inv[i].data_ptr()
        for s in range(len(dX)):
            desc.dseq[s] = dX[s].data_ptr()
            desc.seq_T[s] = seq_T[s]
        desc.dtar = dtar.data_ptr()
        desc.dpooled, desc.ld_pooled = dz.data_ptr(), dz.stride(0)
        desc.grad_dtype = ops.dt_code(dz.dtype)
        grad_rows = self._buf("grad_rows", (prep["cap"], plan.max_dim), F32)
        # only the first n_uniq rows are accumulated into (segment ids are < n_uniq); the count lives on the device
        L.call("dmt_zero_rows", ops.p(grad_rows), ops.p(prep["n_uniq"]), 0, prep["cap"], plan.max_dim, ops.stream_ptr())
        ws, wsb = ops.det_ws(n, plan.max_dim, self.store.device, "embgrad")
        L.call("dmt_embgrad_reduce", C.byref(desc), ops.p(prep["keys_s"]), ops.p(prep["vals_s"]), ops.p(prep["seg"]), n,
               ops.p(grad_rows), plan.max_dim, ws, wsb, ops.stream_ptr())
        self.sparse = (prep["uniq"], prep["n_uniq"], grad_rows, prep["cap"])

    def sort_segments(self, keys, vals, keys_s, vals_s, n, tag="", own=False):
        """Stable sort of (row, entry) pairs + segment ids of equal rows.  `tag` names the scratch set: callers on different streams
        (index plane / compute stream) must not share one.  own: the results are fresh tensors the caller may keep (otherwise views of
        the scratch set, to be cloned before the next call)."""
        store = self.store
        st = ops.stream_ptr()
        end_bit = max(1, int(store.total_rows).bit_length())
        need = C.c_uint64(0)
        L.call("dmt_sort_pairs", ops.p(keys), ops.p(keys_s), ops.p(vals), ops.p(vals_s), n, end_bit, None, C.byref(need), st)
        ws = self._buf(tag + "sort_ws", (max(int(need.value), 16),), torch.uint8)
        have = C.c_uint64(ws.numel())
        L.call("dmt_sort_pairs", ops.p(keys), ops.p(keys_s), ops.p(vals), ops.p(vals_s), n, end_bit, ops.p(ws), C.byref(have), st)
        if own:
            dev = store.device
            seg, uniq, n_uniq = (torch.empty((n,), dtype=torch.int32, device=dev), torch.empty((n,), dtype=torch.int32, device=dev),
                                 torch.empty((1,), dtype=torch.int32, device=dev))
        else:
            seg = self._buf(tag + "seg", (n,), torch.int32)
            uniq = self._buf(tag + "uniq", (n,), torch.int32)
            n_uniq = self._buf(tag + "n_uniq", (1,), torch.int32)
        need2 = C.c_uint64(0)
        L.call("dmt_segment_heads", ops.p(keys_s), n, store.total_rows, ops.p(seg), ops.p(uniq), ops.p(n_uniq), None, C.byref(need2), st)
        ws2 = self._buf(tag + "heads_ws", (max(int(need2.value), 16),), torch.uint8)
        have2 = C.c_uint64(ws2.numel())
        L.call("dmt_segment_heads", ops.p(keys_s), n, store.total_rows, ops.p(seg), ops.p(uniq), ops.p(n_uniq), ops.p(ws2), C.byref(have2), st)
        return uniq, n_uniq, seg

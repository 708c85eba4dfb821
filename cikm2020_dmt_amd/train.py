"""The train step of run_dnn.train()'s tower body (run_dnn.py:148-207), one process per GPU.

    logits = inf.inference(features, is_train=True)            -> DMTEngine.inference
    loss   = inf.loss_multi_task_unbias(logits, labels, mask)  -> DMTEngine.loss_unbias
    grads  = opt.compute_gradients(loss)                       -> loss.backward() (dense arena + sparse rows)
    grads  = average_gradients(tower_grads)                    -> parallel.allreduce_dense_ / merge of sparse rows
    opt.apply_gradients(grads)                                 -> TFAdam.step
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L
from . import ops, parallel
from .engine import DeviceBatch, DMTEngine
from .optim import TFAdam
from .variables import VariableStore


class Trainer:
    def __init__(self, spec: dict, device="cuda", compute_dtype=torch.float32, seed: int = 0, learning_rate=(0.001, 0.0001),
                 step_boundary=(300000000,), init: bool = True, max_steps: int = 1 << 20, dropout: bool = True, dropout_seed: int = 1,
                 dp_exchange: str = "owner", force_dp: bool = False, fused_mhsa=None, table_layout: str = "replicated"):
        """table_layout: "replicated" (every rank holds every embedding table; gradient rows are exchanged and every rank applies the
        same update) or "sharded" (BASELINE configs[3]: rank r holds the rows with id % world == r; ids travel to the owners and rows
        back before the forward pass, gradient rows travel to the owners after the backward pass, only owners run Adam)."""
        self.spec = spec
        self.device = torch.device(device)
        if table_layout not in ("replicated", "sharded"):
            raise ValueError("table_layout must be 'replicated' or 'sharded'")
        self.table_layout = table_layout
        shard = parallel.world() if table_layout == "sharded" else None
        self.store = VariableStore(spec, self.device, compute_dtype, seed=seed, init=init, table_shard=shard)
        self.engine = DMTEngine(spec, self.store)
        self.opt = TFAdam(self.store, learning_rate, step_boundary, max_steps=max_steps)
        self.last = {}
        # is_train semantics of the reference: Transformer dropout 0.1 and bias-tower dropout 0.5 are ALWAYS active in
        # train(), so it is the default here; parity runs against the oracle pass dropout=False (or the same seeds).
        self.dropout, self.dropout_seed = dropout, dropout_seed
        if fused_mhsa is not None:
            self.engine.use_mhsa = bool(fused_mhsa)
        if dp_exchange not in ("owner", "allgather"):
            raise ValueError("dp_exchange must be 'owner' or 'allgather'")
        self.dp_exchange = dp_exchange     # how the embedding-gradient rows cross ranks (parallel.py)
        # run the data-parallel exchange even in a one-rank group (a one-GPU box can still push the real RCCL calls of the
        # N-rank step through a 1-rank communicator: tests/test_gpu_dp.py)
        self.force_dp = bool(force_dp)

    def make_batch(self, inputs, mask=None, label=None, pad_to=None) -> DeviceBatch:
        return DeviceBatch.from_inputs(inputs, self.spec, self.device, mask=mask, label=label, pad_to=pad_to)

    def sync_rows(self, batch: DeviceBatch):
        """Bring the table rows this batch reads up to date (exact lazy Adam), before any kernel gathers them."""
        prep = self.engine.prepare(batch)
        if self.table_layout == "sharded":
            self.engine.fetch_rows(batch, self.opt)      # owners replay the lazy updates of what they send
            return prep
        if self.opt.global_step > 0:
            self.opt.catch_up(prep["uniq"], prep["n_uniq"], prep["cap"])
        return prep

    def forward_backward(self, batch: DeviceBatch):
        self.sync_rows(batch)
        self.store.zero_grad()
        rank, _W = parallel.world()
        self.engine.dropout_step_seed = (self.dropout_seed + self.opt.global_step + 7919 * rank) if self.dropout else None
        out = self.engine.inference(batch)
        loss, p_ctr, p_cvr = self.engine.loss_unbias(out, batch.mask)
        self._early = None
        if _W > 1 or (self.force_dp and parallel.dist.is_initialized()):
            # The gradient arena is laid out [Transformers | MMoE, towers, bias tower].  Everything behind the MMoE input z is
            # final the moment dL/dz exists (91 % of the dense parameters): its all-reduce runs on the collective's own stream
            # while the three Transformer backward passes are still computing (run_dnn.py:45-80 average_gradients).
            z = self.engine.intermediates.get("zbuf")
            off = self.store.leaves["mmoe_layers/l0_cat_weights"].offset
            if z is not None and z.requires_grad:
                def _hook(g, off=off):
                    if self._early is None:
                        self._early = (off, parallel.allreduce_dense_(self.store.grads[off:], async_op=True, force=self.force_dp))
                    return g
                z.register_hook(_hook)
        loss.backward()
        self.engine.dropout_step_seed = None
        self.last = dict(out=out, p_ctr=p_ctr, p_cvr=p_cvr)
        return loss.detach()

    def merge_sparse(self, sparse):
        """average_gradients for the IndexedSlices: concatenate every rank's (row, grad) pairs in rank order and
        reduce rows again with the same stable sort + segment reduce."""
        rank, W = parallel.world()
        if W == 1 and not (self.force_dp and parallel.dist.is_initialized()):
            return sparse                      # (one rank owns every row: sharded == replicated)
        uniq, n_uniq, grad_rows, _cap = sparse
        n = int(n_uniq.item())
        eng, st = self.engine, self.store
        wire = torch.bfloat16 if st.compute_dtype == torch.bfloat16 else None    # bf16 mode: gradient rows travel as bf16
        if self.table_layout == "sharded":
            # every (row, gradient row) pair goes to the row's owner, which reduces them in rank order and is the only one to apply Adam
            rk, rr = parallel.exchange_to_owners(uniq, grad_rows, n, transport_dtype=wire, group_fn=self._group_by_owner)
            if rk.numel() > 0:
                return self.merge_gathered(rk, rr)
            return (uniq[:0], torch.zeros(1, dtype=torch.int32, device=uniq.device), grad_rows[:0], 0)
        if self.dp_exchange == "allgather":
            all_k, all_r, cap = parallel.allgather_sparse(uniq, grad_rows, n, st.total_rows, transport_dtype=wire)
            return self.merge_gathered(all_k, all_r)
        # owner-reduce: a row's contributions meet on rank row % W (1/W of the pairs per rank instead of all of them on every
        # rank), are reduced there in rank order, and only the REDUCED shards are all-gathered: at 8 ranks 456 MB instead of
        # 827 MB received per rank and a 1/8 second-level reduce
        rk, rr = parallel.exchange_to_owners(uniq, grad_rows, n, transport_dtype=wire, group_fn=self._group_by_owner)
        if rk.numel() > 0:
            uniq2, m, shard_rows, _capm = self.merge_gathered(rk, rr)       # m stays on the device: allgather_shards syncs once
        else:
            uniq2, shard_rows, m = uniq[:0], grad_rows[:0], 0
        all_k, all_r, cap = parallel.allgather_shards(uniq2, shard_rows, m, st.total_rows, transport_dtype=wire)
        N = all_k.numel()
        n_dev = torch.full((1,), N, dtype=torch.int32, device=all_k.device)
        return (all_k, n_dev, all_r, N)

    def _group_by_owner(self, keys, owner, rows, wire):
        """Stable grouping of this rank's (key, row) pairs by owner rank: ONE radix pass over the owner ids (they fit in a few
        bits) and one gather kernel that also rounds the rows to the wire format."""
        eng = self.engine
        n = keys.numel()
        _r, W = parallel.world()
        end_bit = max(1, int(W - 1).bit_length())
        iota = torch.arange(n, dtype=torch.int32, device=keys.device)
        own_s = eng._buf("own_keys_s", (n,), torch.int32)
        perm32 = eng._buf("own_perm", (n,), torch.int32)
        own = owner.to(torch.int32) if owner.dtype != torch.int32 else owner
        need = C.c_uint64(0)
        L.call("dmt_sort_pairs", ops.p(own), ops.p(own_s), ops.p(iota), ops.p(perm32), n, end_bit, None, C.byref(need), ops.stream_ptr())
        ws = eng._buf("own_sort_ws", (max(int(need.value), 16),), torch.uint8)
        have = C.c_uint64(ws.numel())
        L.call("dmt_sort_pairs", ops.p(own), ops.p(own_s), ops.p(iota), ops.p(perm32), n, end_bit, ops.p(ws), C.byref(have), ops.stream_ptr())
        perm = perm32.long()
        send_k = keys.index_select(0, perm)
        out_dt = wire if wire is not None else rows.dtype
        send_r = torch.empty((n, rows.shape[1]), dtype=out_dt, device=rows.device)
        if rows.dtype == torch.float32 and rows.is_contiguous() and rows.shape[1] % 4 == 0 and out_dt in (torch.float32, torch.bfloat16):
            L.call("dmt_rows_permute", ops.p(rows), ops.p(perm), n, int(rows.shape[1]), ops.dt_code(out_dt), ops.p(send_r), ops.stream_ptr())
        else:
            send_r.copy_(rows.index_select(0, perm))
        return send_k, send_r

    def merge_gathered(self, all_k, all_r):
        """Second-level reduce of the rank-major (row id, gradient row) pairs: same stable sort + segment reduce as the
        per-rank embedding gradient (also driven directly by scripts/dp_merge_bench.py with a synthetic 8-rank gather)."""
        eng, st = self.engine, self.store
        grad_rows = all_r
        N = all_k.numel()
        vals = torch.arange(N, dtype=torch.int32, device=all_k.device)
        keys_s = eng._buf("m_keys_s", (N,), torch.int32)
        vals_s = eng._buf("m_vals_s", (N,), torch.int32)
        uniq2, n_uniq2, seg = eng.sort_segments(all_k, vals, keys_s, vals_s, N)
        uniq2, n_uniq2 = uniq2.clone(), n_uniq2.clone()
        capm = min(N, st.total_rows)
        out_rows = eng._buf("m_rows", (capm, grad_rows.shape[1]), torch.float32)
        L.call("dmt_zero_rows", ops.p(out_rows), ops.p(n_uniq2), 0, capm, int(grad_rows.shape[1]), ops.stream_ptr())
        L.call("dmt_rows_reduce_bf16" if all_r.dtype == torch.bfloat16 else "dmt_rows_reduce", ops.p(keys_s), ops.p(vals_s), ops.p(seg), N,
               st.total_rows, ops.p(all_r), ops.p(out_rows), int(grad_rows.shape[1]), ops.stream_ptr())
        return (uniq2, n_uniq2, out_rows, capm)

    def train_step(self, batch: DeviceBatch):
        loss = self.forward_backward(batch)
        rank, W = parallel.world()
        sparse = self.engine.sparse
        if W > 1 or (self.force_dp and parallel.dist.is_initialized()):
            early, self._early = getattr(self, "_early", None), None
            self.early_allreduce_used = early is not None
            if early is not None:
                works = [early[1], parallel.allreduce_dense_(self.store.grads[: early[0]], async_op=True, force=self.force_dp)]
            else:
                works = [parallel.allreduce_dense_(self.store.grads, async_op=True, force=self.force_dp)]
            sparse = self.merge_sparse(sparse)
            for w in works:
                if w is not None:
                    w.wait()
            loss = parallel.mean_scalar(loss)
        self.opt.step(sparse, grad_scale=1.0 / W)
        return loss

    @torch.no_grad()
    def predict(self, batch: DeviceBatch):
        """run_dnn.predict scoring (run_dnn.py:663-687): sigmoid(logit + y_bias)."""
        self.sync_rows(batch)
        with torch.enable_grad():
            out = self.engine.inference(batch)
        (c, o), yb = out
        return torch.sigmoid(c + yb).detach(), torch.sigmoid(o + yb).detach()

"""The train step of run_dnn.train()'s tower body (run_dnn.py:148-207), one process per GPU.

    logits = inf.inference(features, is_train=True)            -> DMTEngine.inference
    loss   = inf.loss_multi_task_unbias(logits, labels, mask)  -> DMTEngine.loss_unbias
    grads  = opt.compute_gradients(loss)                       -> loss.backward() (dense arena + sparse rows)
    grads  = average_gradients(tower_grads)                    -> parallel.allreduce_dense_ / merge of sparse rows
    opt.apply_gradients(grads)                                 -> TFAdam.step
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import torch

from . import _lib as L
from . import ops, parallel, streams
from .engine import DeviceBatch, DMTEngine
from .optim import make_optimizer
from .variables import VariableStore


def _record_stream(obj, stream):
    """Tensors made on the index stream and consumed on the compute stream: tell the caching allocator (it would otherwise hand a
    freed block back to the index stream while the compute stream -- which the host runs far ahead of -- still reads it)."""
    if isinstance(obj, torch.Tensor):
        if obj.is_cuda:
            obj.record_stream(stream)
    elif isinstance(obj, dict):
        for v in obj.values():
            _record_stream(v, stream)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            _record_stream(v, stream)


class Trainer:
    def __init__(self, spec: dict, device="cuda", compute_dtype=torch.float32, seed: int = 0, learning_rate=(0.001, 0.0001),
                 step_boundary=(300000000,), init: bool = True, max_steps: int = 1 << 20, dropout: bool = True, dropout_seed: int = 1,
                 dp_exchange: str = "owner", force_dp: bool = False, fused_mhsa=None, table_layout: str = "replicated", attn_dtype=None,
                 wgrad320_min_rows=None, packed_rows=None, optimizer: str = "adam", wnd_wd: float = 0.0, l2_emb_lambda: float = 0.01,
                 l2_batch_size=None):
        """table_layout: "replicated" (every rank holds every embedding table; gradient rows are exchanged and every rank applies the
        same update) or "sharded" (BASELINE configs[3]: rank r holds the rows with id % world == r; ids travel to the owners and rows
        back before the forward pass, gradient rows travel to the owners after the backward pass, only owners run Adam)."""
        self.spec = spec
        self.device = torch.device(device)
        if attn_dtype not in (None, "bf16", "fp8"):
            raise ValueError("attn_dtype must be None, 'bf16' or 'fp8'")
        if table_layout not in ("replicated", "sharded"):
            raise ValueError("table_layout must be 'replicated' or 'sharded'")
        self.table_layout = table_layout
        shard = parallel.world() if table_layout == "sharded" else None
        self.store = VariableStore(spec, self.device, compute_dtype, seed=seed, init=init, table_shard=shard)
        self.engine = DMTEngine(spec, self.store)
        # what a step in flight keeps (collected weight gradients, their lane, the long-row threshold) belongs to THIS trainer's engine
        self.engine.step_state = ops.StepState(wgrad320_min_rows)
        # BASELINE configs[4]: the long-sequence (64 < T <= 256) attention forward of THIS trainer multiplies in OCP e4m3
        self.engine.kopts = ops.KernelOptions(attn_mma_fp8=(attn_dtype == "fp8"))
        # get_optimizer(optimizer, learning_rate) (model/inference_mlp.py:264-280): adam (dmt.conf:70) or one of the other five
        self.opt = make_optimizer(optimizer, self.store, learning_rate, step_boundary, max_steps=max_steps)
        # run_dnn.py:174-175: tower_train_loss += inf.l2_norm(features) when wnd_wd > 1e-5 (dmt.conf: 0.0); the term is scaled by
        # l2_emb_lambda / batch_size of the conf (mmoe_transformer_unbias.py:58-59; l2_batch_size None: the batch at hand)
        self.wnd_wd, self.l2_emb_lambda, self.l2_batch_size = float(wnd_wd), float(l2_emb_lambda), l2_batch_size
        self.last = {}
        self.diag = None       # dict: train_step brackets its phases with HIP events (key -> [(start, end)]); bench.py's step_phases_ms
        # is_train semantics of the reference: Transformer dropout 0.1 and bias-tower dropout 0.5 are ALWAYS active in
        # train(), so it is the default here; parity runs against the oracle pass dropout=False (or the same seeds).
        self.dropout, self.dropout_seed = dropout, dropout_seed
        if fused_mhsa is not None:
            self.engine.use_mhsa = bool(fused_mhsa)
        self.batch_dw = os.environ.get("DMT_BATCH_DW", "1") == "1"      # B-row weight gradients of the one-GPU backward: one launch per lane
        if packed_rows is not None:          # (default: on, DMT_PACKED_ROWS; engine.DMTEngine.seq_pack decides per batch and sequence)
            self.engine.packed_rows = bool(packed_rows)
        if dp_exchange not in ("owner", "allgather"):
            raise ValueError("dp_exchange must be 'owner' or 'allgather'")
        self.dp_exchange = dp_exchange     # how the embedding-gradient rows cross ranks (parallel.py)
        # run the data-parallel exchange even in a one-rank group (a one-GPU box can still push the real RCCL calls of the
        # N-rank step through a 1-rank communicator: tests/test_gpu_dp.py)
        self.force_dp = bool(force_dp)
        if os.environ.get("DMT_DETERMINISTIC") == "1" and not ops.DETERMINISTIC:
            ops.set_deterministic(True)
        # data-parallel step: long-row weight gradients collected in backward and launched beside the row exchange (train_step).  On
        # by default with more than one rank; in a ONE-rank group (no wire time to hide) it only costs the overlap the lanes give
        # those kernels inside backward (10.8 against 10.4 ms), so there it is off unless asked for
        ov = os.environ.get("DMT_DP_OVERLAP_WGRADS", "auto")
        self.overlap_wgrads = (parallel.world()[1] > 1) if ov == "auto" else (ov == "1")
        self.fork_mmoe_wgrads = os.environ.get("DMT_FORK_MMOE_WGRADS", "1") == "1"    # one-GPU backward: MMoE / tower weight gradients on an idle lane
        self.sparse_lane = os.environ.get("DMT_SPARSE_LANE", "0") == "1"     # one-GPU step: id-bound tail beside the deferred weight gradients (off: even at L=50, -3 % at L=200)
        self.index_stream, self._ix_stream = os.environ.get("DMT_INDEX_STREAM", "1") == "1", None    # index plane (id sort, exchange plan) on a side stream: sync_rows
        # lazy Adam: the pending zero-gradient updates of batch i + 1's rows can be replayed on the index lane WHILE step i runs
        # (train_step(prefetch=)) instead of in front of step i + 1's gather.  Exact (tests/test_gpu_boundary.py), but OFF by default:
        # the one-GPU step is throughput-bound, not latency-bound -- measured 10.12 ms with it against 10.01 ms without (the replay
        # takes the same GPU time on the index lane, 0.83 ms, and slows the kernels it runs beside)
        self.early_catchup = os.environ.get("DMT_EARLY_CATCHUP", "0") == "1"
        self.early_catchup_at_junction = os.environ.get("DMT_EARLY_CATCHUP_AT", "junction") == "junction"
        self._inflight = None        # (event "begin() of the step in flight has run and its rows are stamped", that step's local number)
        if self.device.type == "cuda":
            streams.lanes(self.device)                     # bind the step's lanes to hardware queues before anything else (streams.py)
            if self._dp_active():
                streams.warm_communicators(self.device)

    class _Span:
        """with self._span("key"): ...  -- a HIP-event pair on the CURRENT stream around the block when self.diag is a dict (else free)."""
        __slots__ = ("tr", "key", "e0")

        def __init__(self, tr, key):
            self.tr, self.key, self.e0 = tr, key, None

        def __enter__(self):
            if self.tr.diag is not None and self.tr.device.type == "cuda":
                self.e0 = torch.cuda.Event(enable_timing=True)
                self.e0.record()
            return self

        def __exit__(self, *a):
            if self.e0 is not None:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record()
                self.tr.diag.setdefault(self.key, []).append((self.e0, e1))
            return False

    def _span(self, key):
        return Trainer._Span(self, key)

    def _mark(self, key, e0=None):
        """Open (e0 None -> returns the start event) or close a span whose two ends lie in different methods."""
        if self.diag is None or self.device.type != "cuda":
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        if e0 is not None:
            self.diag.setdefault(key, []).append((e0, ev))
        return ev

    def make_batch(self, inputs, mask=None, label=None, pad_to=None) -> DeviceBatch:
        return DeviceBatch.from_inputs(inputs, self.spec, self.device, mask=mask, label=label, pad_to=pad_to)

    def _needs_plan(self, for_training):
        return (for_training or self.table_layout == "sharded") and self._dp_active() and self.dp_exchange == "owner"

    def _index_plane(self, batch: DeviceBatch, need_plan: bool):
        """engine.prepare (+ plan_exchange) for a batch.  Everything here depends on the batch's ids only -- not on any step's results --
        so it runs on its own stream, next to whatever the compute stream has queued, and its host syncs wait for THIS stream only:
        the compute stream never drains.  prep["_ready"] is the event the compute stream waits for before it touches the result."""
        side = self._index_stream()
        if side is None:
            prep = self.engine.prepare(batch)
            if need_plan and "xplan" not in prep:
                prep["xplan"] = self.plan_exchange(prep["uniq"], prep["n_uniq"])
            return prep
        main = ops.cur_stream(self.device)
        if batch.ready is not None:
            side.wait_event(batch.ready)
            self._record_batch(batch, main)      # (the id columns are read on this lane, possibly before forward_backward sees the batch)
        else:
            side.wait_stream(main)
        with torch.cuda.stream(side):
            with self._span("index_plane_sort"):
                prep = self.engine.prepare(batch)
            if need_plan and "xplan" not in prep:
                with self._span("exchange_ids"):
                    prep["xplan"] = self.plan_exchange(prep["uniq"], prep["n_uniq"])
            _record_stream(prep, main)
            ev = torch.cuda.Event()
            ev.record(side)
        prep["_ready"] = ev
        return prep

    def prefetch(self, batch: DeviceBatch, for_training: bool = True):
        """Start the index plane of a LATER batch now (call it with batch i + 1 while step i is being issued: train_step(prefetch=)).
        In a data-parallel step the id exchange and its host syncs then overlap step i instead of opening step i + 1."""
        need_plan = self._needs_plan(for_training)
        prep = getattr(batch, "_prep", None)
        if prep is None or (need_plan and "xplan" not in prep):
            prep = self._index_plane(batch, need_plan)
        self._catch_up_early(prep)

    def _catch_up_early(self, prep):
        """Replay, on the index lane and while the current step is still running, the pending zero-gradient Adam updates of the rows a
        LATER batch reads -- through the step in flight, for every row that step does not update itself (stamped rows: it brings
        those up to date when it applies their gradient).  Exact: a zero-gradient update of step t needs (p, m, v) and lr_t only, and
        lr_t is in the history since that step's begin().  Off the critical path: the replay (0.9 ms with real id gaps) used to sit
        between the optimizer of one step and the gather of the next."""
        infl, side = self._inflight, self._index_stream()
        if infl is None or side is None or prep is None or "_caught" in prep or self.table_layout == "sharded":
            return
        ev_begun, to_step = infl
        main = ops.cur_stream(self.device)
        with torch.cuda.stream(side):
            side.wait_event(ev_begun)
            jev = getattr(self.engine, "junction_event", None)
            if jev is not None and self.early_catchup_at_junction:
                # not beside the forward's long kernels (measured: no gain, the replay takes its issue slots from them) but where the
                # step leaves the chip nearly idle: the chain of B-row kernels between the last decoder and the first long backward kernel
                side.wait_event(jev)
            with self._span("adam_catchup_early"):
                self.opt.catch_up_early(prep["uniq"], prep["n_uniq"], prep["cap"], to_step)
            ev = torch.cuda.Event()
            ev.record(side)
        prep["_caught"] = ev
        _record_stream(prep, main)

    def sync_rows(self, batch: DeviceBatch, for_training: bool = True):
        """Bring the table rows this batch reads up to date (exact lazy Adam), before any kernel gathers them; in a data-parallel
        training step also the index plane of the gradient exchange (plan_exchange), unless prefetch() already ran it."""
        need_plan = self._needs_plan(for_training)
        prep = getattr(batch, "_prep", None)
        if prep is None or (need_plan and "xplan" not in prep):
            prep = self._index_plane(batch, need_plan)
        for key in ("_ready", "_caught"):
            ev = prep.pop(key, None)
            if ev is not None:
                ops.cur_stream(self.device).wait_event(ev)
        if self.table_layout == "sharded":
            with self._span("row_fetch"):
                self.engine.fetch_rows(batch, self.opt, prep.get("xplan"))      # owners replay the lazy updates of what they send
            return prep
        if self.opt.global_step > 0:
            with self._span("adam_catchup"):
                self.opt.catch_up(prep["uniq"], prep["n_uniq"], prep["cap"])
        return prep

    def _record_batch(self, batch: DeviceBatch, cur):
        """Every tensor of a batch uploaded on another stream, recorded once on every stream of the step that reads it: the compute
        stream, the index lane (embgrad keys / sort of the id columns) and the sequence lanes (lens).  On the packed path all columns
        are views of one storage; on the column-by-column path (parser.pinned = False, from_inputs under a side stream) each column is
        its own allocation, and an unrecorded one could be handed to the next upload while this step still reads it."""
        if getattr(batch, "_recorded_for", None) is self:
            return
        lanes = [cur]
        side = self._index_stream()
        if side is not None:
            lanes.append(side)
        if self.engine.seq_streams and self.device.type == "cuda":
            n_seq = len(self.spec["attention_embed_pairs"])
            lanes += [st for st in self.engine._seq_stream_pool(max(2, n_seq)) if st is not None]
        seen = set()
        tensors = [batch.dense, batch.mask, batch.label]
        for col in batch.feats.values():
            tensors += [col.idx, col.wts, col.lens]
        for t in tensors:
            if t is None or not t.is_cuda:
                continue
            key = t.untyped_storage().data_ptr()
            if key in seen:
                continue
            seen.add(key)
            for st in lanes:
                t.record_stream(st)
        batch._recorded_for = self

    def _index_stream(self):
        if self.device.type != "cuda" or not self.index_stream:
            return None
        if self._ix_stream is None:
            self._ix_stream = streams.lanes(self.device)["index"]
        return self._ix_stream

    def _dp_active(self):
        return parallel.world()[1] > 1 or (self.force_dp and parallel.dist.is_initialized())

    def plan_exchange(self, uniq, n_uniq):
        """INDEX PLANE of the embedding-gradient exchange, run before the forward pass.  Which rows this rank will hold gradients for
        is known as soon as the batch's ids are sorted (engine.prepare), so everything that depends only on ids happens here, off the
        critical path between backward and the optimizer: owner grouping of the distinct rows (row % W), the all_to_all of the row
        ids, the owner's stable sort + segmentation of what it received (rank order = the order of summation of run_dnn.py:45-80) and --
        replicated layout -- the all-gather of the owners' distinct-row lists.  The two host syncs this needs (pair counts per
        (sender, owner); distinct rows per owner) therefore happen while the device queue is still short; after backward only gradient
        ROWS move (exchange_rows_begin / _finish), with every size already on the host."""
        eng, st = self.engine, self.store
        rank, W = parallel.world()
        dev = uniq.device
        cap = uniq.numel()
        iota = torch.arange(cap, dtype=torch.int32, device=dev)
        owner = torch.where(iota < n_uniq, torch.remainder(uniq, W), W).to(torch.int32)       # slots past n_uniq -> bucket W (sorted last)
        # stable grouping by owner: one radix pass over the few owner bits; the pair counts per owner fall out of the sorted owner ids
        own_s = torch.empty((cap,), dtype=torch.int32, device=dev)
        perm32 = torch.empty((cap,), dtype=torch.int32, device=dev)
        end_bit = max(1, int(W).bit_length())
        need = C.c_uint64(0)
        L.call("dmt_sort_pairs", ops.p(owner), ops.p(own_s), ops.p(iota), ops.p(perm32), cap, end_bit, None, C.byref(need), ops.stream_ptr())
        ws = eng._buf("ix_own_sort_ws", (max(int(need.value), 16),), torch.uint8)
        have = C.c_uint64(ws.numel())
        L.call("dmt_sort_pairs", ops.p(owner), ops.p(own_s), ops.p(iota), ops.p(perm32), cap, end_bit, ops.p(ws), C.byref(have), ops.stream_ptr())
        edges = torch.searchsorted(own_s, torch.arange(W + 1, dtype=torch.int32, device=dev))
        counts = edges[1:] - edges[:-1]
        grp = parallel.index_group() if parallel.dist.is_initialized() else None
        if parallel.dist.is_initialized():
            mat = [torch.zeros_like(counts) for _ in range(W)]
            parallel.dist.all_gather(mat, counts, group=grp)
            M = torch.stack(mat).cpu()                                    # host sync 1 of 2 (before the forward pass)
        else:
            M = counts.cpu().reshape(1, 1)
        send_splits, recv_splits = M[rank].tolist(), M[:, rank].tolist()
        n, Rn = int(sum(send_splits)), int(sum(recv_splits))
        perm = perm32[:n].long()
        send_k = uniq.index_select(0, perm)
        recv_k = torch.empty((Rn,), dtype=uniq.dtype, device=dev)
        if parallel.dist.is_initialized():
            parallel._a2a(recv_k, send_k, recv_splits, send_splits, group=grp)
        else:
            recv_k.copy_(send_k)
        plan = dict(n=n, R=Rn, perm=perm, send_splits=send_splits, recv_splits=recv_splits, recv_k=recv_k)
        if Rn > 0:
            vals = torch.arange(Rn, dtype=torch.int32, device=dev)
            keys_s = torch.empty((Rn,), dtype=torch.int32, device=dev)
            vals_s = torch.empty((Rn,), dtype=torch.int32, device=dev)
            uniq2, n_uniq2, seg = eng.sort_segments(recv_k, vals, keys_s, vals_s, Rn, own=True)
            plan.update(keys_s=keys_s, vals_s=vals_s, seg=seg, uniq2=uniq2[:Rn], n_uniq2=n_uniq2)
        else:
            plan.update(keys_s=None, uniq2=uniq[:0], n_uniq2=torch.zeros(1, dtype=torch.int32, device=dev))
        if self.table_layout == "sharded":
            return plan
        # replicated layout: every rank applies every owner's reduced rows -> the owners' distinct-row lists are all-gathered now
        cnt = plan["n_uniq2"].to(torch.int64)
        if parallel.dist.is_initialized():
            cnts = [torch.zeros_like(cnt) for _ in range(W)]
            parallel.dist.all_gather(cnts, cnt, group=grp)
            c_host = torch.stack(cnts).reshape(-1).cpu()                  # host sync 2 of 2 (still before the forward pass)
        else:
            c_host = cnt.cpu()
        m = int(c_host[rank])
        cap_m = max(1, int(c_host.max()))
        k_loc = torch.full((cap_m,), st.total_rows, dtype=uniq.dtype, device=dev)
        k_loc[:m] = plan["uniq2"][:m]
        all_k = torch.empty((W * cap_m,), dtype=uniq.dtype, device=dev)
        if parallel.dist.is_initialized():
            parallel._all_gather_cat(all_k, k_loc, W, cap_m, group=grp)
        else:
            all_k.copy_(k_loc)
        plan.update(m=m, cap_m=cap_m, all_k=all_k, n_dev=torch.full((1,), W * cap_m, dtype=torch.int32, device=dev))
        return plan

    def exchange_rows_begin(self, sparse, plan):
        """DATA PLANE, first half (right after backward): this rank's gradient rows, grouped by owner and rounded to the wire format by one
        kernel (dmt_rows_permute), leave in one all_to_all_single.  With RCCL the collective is asynchronous: the dense Adam step runs
        while the rows are on the links."""
        _uniq, _n_uniq, grad_rows, _cap = sparse
        st = self.store
        wire = torch.bfloat16 if st.compute_dtype == torch.bfloat16 else torch.float32
        n, Rn, D = plan["n"], plan["R"], int(grad_rows.shape[1])
        send_r = torch.empty((n, D), dtype=wire, device=grad_rows.device)
        if n > 0:
            if grad_rows.dtype == torch.float32 and grad_rows.is_contiguous() and D % 4 == 0:
                L.call("dmt_rows_permute", ops.p(grad_rows), ops.p(plan["perm"]), n, D, ops.dt_code(wire), ops.p(send_r), ops.stream_ptr())
            else:
                send_r.copy_(grad_rows.index_select(0, plan["perm"]))
        recv_r = torch.empty((Rn, D), dtype=wire, device=grad_rows.device)
        work = None
        if not parallel.dist.is_initialized():
            recv_r.copy_(send_r)
        elif parallel.dist.get_backend() == "nccl":
            work = parallel.dist.all_to_all_single(recv_r, send_r, plan["recv_splits"], plan["send_splits"], async_op=True)
        else:
            parallel._a2a(recv_r, send_r, plan["recv_splits"], plan["send_splits"])
        return (work, send_r, recv_r)

    def exchange_rows_reduce(self, handle, plan, t_issue=None):
        """DATA PLANE, second part: the owner sums what it received per row, in rank order, with the segments prepared by
        plan_exchange (dmt_rows_reduce).  Sharded layout: done -- the owner applies Adam to its rows.  Replicated layout: the reduced
        shards leave in an all-gather (padded to the largest shard; padding slots carry an invalid key the optimizer kernels skip),
        asynchronous with RCCL: exchange_rows_collect waits for it."""
        work, _send_r, recv_r = handle
        if work is not None:
            work.wait()
        self._mark("exchange_rows_a2a", t_issue)          # issue -> the compute stream may read what arrived
        eng, st = self.engine, self.store
        rank, W = parallel.world()
        Rn, D = plan["R"], int(recv_r.shape[1])
        sharded = self.table_layout == "sharded"
        rows_cap = max(1, Rn if sharded else max(plan["cap_m"], Rn))
        out_rows = eng._buf("m_rows", (rows_cap, D), torch.float32)
        if Rn > 0:
            L.call("dmt_zero_rows", ops.p(out_rows), ops.p(plan["n_uniq2"]), 0, rows_cap, D, ops.stream_ptr())
            ws, wsb = ops.det_ws(Rn, D, recv_r.device, "rows")
            L.call("dmt_rows_reduce_bf16" if recv_r.dtype == torch.bfloat16 else "dmt_rows_reduce", ops.p(plan["keys_s"]), ops.p(plan["vals_s"]),
                   ops.p(plan["seg"]), Rn, st.total_rows, ops.p(recv_r), ops.p(out_rows), D, ws, wsb, ops.stream_ptr())
        if sharded:
            return (None, (plan["uniq2"], plan["n_uniq2"], out_rows, Rn), None, None)
        cap_m = plan["cap_m"]
        r_loc = out_rows[:cap_m]
        if recv_r.dtype == torch.bfloat16:
            r_loc = r_loc.to(torch.bfloat16)          # (rows past this shard's m are never read: their keys are invalid)
        r_loc = r_loc.contiguous()
        all_r = torch.empty((W * cap_m, D), dtype=r_loc.dtype, device=r_loc.device)
        work2 = None
        t_ag = self._mark("exchange_rows_allgather")
        if not parallel.dist.is_initialized():
            all_r.copy_(r_loc)
        elif parallel.dist.get_backend() == "nccl" and hasattr(parallel.dist, "all_gather_into_tensor"):
            work2 = parallel.dist.all_gather_into_tensor(all_r, r_loc, async_op=True)
        else:
            parallel._all_gather_cat(all_r, r_loc, W, cap_m)
        return (work2, (plan["all_k"], plan["n_dev"], all_r, W * cap_m), r_loc, t_ag)

    def exchange_rows_collect(self, handle2):
        work2, sparse, _keep, t_ag = handle2
        if work2 is not None:
            work2.wait()
        if t_ag is not None:
            self._mark("exchange_rows_allgather", t_ag)
        return sparse

    def exchange_rows_finish(self, handle, plan):
        """exchange_rows_reduce + exchange_rows_collect in one go."""
        return self.exchange_rows_collect(self.exchange_rows_reduce(handle, plan))

    def _open_step(self, batch):
        """train_step only: begin() of the optimizer step NOW (lr_t into the history, device step counter advanced) and stamp the rows
        this step will update, so that the index lane may catch the next batch's other rows up through this step (prefetch)."""
        self._inflight = None
        if not self.early_catchup or self.table_layout == "sharded" or self._index_stream() is None:
            return
        prep = getattr(batch, "_prep", None)
        if prep is None:
            return
        dp = self._dp_active()
        plan = prep.get("xplan")
        if dp and (self.dp_exchange != "owner" or plan is None or "all_k" not in plan):
            return                       # (the one-shot exchange forms learn the union of the ranks' rows only after backward)
        self.opt.begin()
        if dp:
            self.opt.stamp_rows(plan["all_k"], plan["n_dev"], int(plan["all_k"].numel()))     # every rank's rows: each replica applies them all
        else:
            self.opt.stamp_rows(prep["uniq"], prep["n_uniq"], prep["cap"])
        ev = torch.cuda.Event()
        ev.record(ops.cur_stream(self.device))
        self._inflight = (ev, self.opt.step_in_flight())

    def _abort_open_step(self):
        """A step that raised after _open_step() had begun the optimizer step early: roll the optimizer back (TFAdam.abort_step), or
        the next step would replay rows through a step that was never applied and then apply the same step number again."""
        self._inflight = None
        try:
            self.opt.abort_step()
        except Exception:          # (the device itself is gone: the original exception is the one to report)
            pass

    def forward_backward(self, batch: DeviceBatch, join: bool = True, prefetch: DeviceBatch = None, defer_wgrads: bool = False,
                         open_step: bool = False):
        """join=False (one-GPU train_step with the sparse lane): backward only collects the long-row weight gradients and leaves the
        embedding-gradient tail to the caller.  prefetch: the NEXT batch -- its index plane is issued once this step's forward is queued."""
        ops.activate(self.engine.step_state)
        ops.reset_deferred_wgrads()       # (closures a step that raised half-way left behind reference that step's tensors)
        self.engine._pending_sparse = None
        if batch.ready is not None and self.device.type == "cuda":
            # a batch uploaded on another stream (an input thread's): the compute stream waits for the copy, and the allocator learns
            # that this stream reads the buffer (a no-op for a batch made on this stream)
            cur = ops.cur_stream(self.device)
            cur.wait_event(batch.ready)
            self._record_batch(batch, cur)
        self.sync_rows(batch)
        if open_step:
            self._open_step(batch)
        self.store.zero_grad()
        rank, _W = parallel.world()
        self.engine.dropout_step_seed = (self.dropout_seed + self.opt.global_step + 7919 * rank) if self.dropout else None
        out = self.engine.inference(batch)
        loss, p_ctr, p_cvr = self.engine.loss_unbias(out, batch.mask)
        if self.wnd_wd > 1e-5:
            loss = loss + self.engine.l2_norm(batch, self.l2_emb_lambda / float(self.l2_batch_size or batch.B))
        if prefetch is not None and prefetch is not batch:
            self.prefetch(prefetch)
        defer = (not join) and self.sparse_lane and self.device.type == "cuda" and self._index_stream() is not None
        dp = _W > 1 or (self.force_dp and parallel.dist.is_initialized())
        if defer:
            ops.begin_deferred_wgrads()
            self.engine.defer_sparse = True
        elif defer_wgrads:
            # (data-parallel step: the caller launches them while the gradient rows are on the links -- train_step.)  The arena's tail
            # is all-reduced from the dL/dz hook below while backward is still running: only weight gradients of the head (the
            # Transformers) may be collected -- with a per-rank batch >= WGRAD320_MIN_ROWS the MMoE layer 0 and the towers' hidden
            # layers qualify by their row count, and collecting them would leave them out of the all-reduce
            ops.begin_deferred_wgrads(limit=self.store.leaves["mmoe_layers/l0_cat_weights"].offset if dp else None)
        self._early = None
        if dp:
            # The gradient arena is laid out [Transformers | MMoE, towers, bias tower].  Everything behind the MMoE input z is
            # final the moment dL/dz exists (91 % of the dense parameters): its all-reduce runs on the collective's own stream
            # while the three Transformer backward passes are still computing (run_dnn.py:45-80 average_gradients).
            z = self.engine.intermediates.get("zbuf")
            off = self.store.leaves["mmoe_layers/l0_cat_weights"].offset
            if z is not None and z.requires_grad:
                def _hook(g, off=off):
                    if self._early is None:
                        self._early_t0 = self._mark("allreduce_tail")
                        self._early = (off, parallel.allreduce_dense_(self.store.grads[off:], async_op=True, force=self.force_dp))
                    return g
                z.register_hook(_hook)
        fork_lane = None
        if (not dp) and self.fork_mmoe_wgrads and self.device.type == "cuda" and self.engine.seq_streams:
            pool = [st for st in self.engine._seq_stream_pool(max(2, len(self.spec["attention_embed_pairs"]))) if st is not None]
            if pool:
                fork_lane = pool[-1]
                ops.begin_fork_wgrads(fork_lane, self.store.leaves["mmoe_layers/l0_cat_weights"].offset)
        # one-GPU step: the B-row weight gradients (decoders, MMoE, towers: 25 launches of 13-17 us that accumulate into the gradient arena
        # and that nothing reads before the optimizer) are collected per lane and leave in one launch each (ops.flush_dw_batches below).
        # In a data-parallel step the arena's tail is all-reduced from the dL/dz hook while backward runs: they stay in place there.
        batch_dw = (not dp) and self.batch_dw and self.device.type == "cuda"
        if batch_dw:
            ops.begin_dw_batching()
            ops.begin_ln_finish_batching()         # (the twelve dgamma / dbeta reductions of the step: one launch behind backward)
        try:
            # d loss / d loss = 1: a cached scalar instead of the ones_like() fill autograd would launch, and the loss kernel's saved
            # logit gradients are handed on as they are (ops.StepState.unit_loss_grad) instead of through a `* 1` launch
            one = getattr(self, "_unit_grad", None)
            if one is None or one.device != loss.device or one.dtype != loss.dtype:
                one = self._unit_grad = torch.ones((), dtype=loss.dtype, device=loss.device)
            self.engine.step_state.unit_loss_grad = True
            try:
                loss.backward(gradient=one)
            finally:
                self.engine.step_state.unit_loss_grad = False
        except BaseException:
            ops.reset_deferred_wgrads()
            self.engine._pending_sparse = None
            self._abort_open_step()
            raise
        finally:
            self.engine.defer_sparse = False
            forked = ops.end_fork_wgrads()
            if batch_dw and sys.exc_info()[0] is not None:
                self.engine.step_state.dw_batch = None         # (a step that raised: drop what it collected)
                self.engine.step_state.ln_finish = None
        if batch_dw:
            ops.flush_ln_finish(end=True)
            cur = ops.cur_stream(self.device)
            for st in ops.flush_dw_batches(end=True):
                if st != cur:
                    cur.wait_stream(st)                         # the batched weight gradients are part of this backward
        if fork_lane is not None and forked is not None and forked["n"]:
            ops.cur_stream(self.device).wait_stream(fork_lane)     # the forked weight gradients are part of this backward
        self.n_forked = forked["n"] if forked is not None else 0
        self.engine.dropout_step_seed = None
        self.last = dict(out=out, p_ctr=p_ctr, p_cvr=p_cvr)
        return loss.detach()

    def merge_sparse(self, sparse):
        """average_gradients for the IndexedSlices: concatenate every rank's (row, grad) pairs in rank order and
        reduce rows again with the same stable sort + segment reduce.  (One-shot form with its host syncs after backward: used by
        dp_exchange="allgather" and when no exchange plan was made -- forward_backward called without sync_rows' plan.)"""
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
        uniq2, n_uniq2, seg = eng.sort_segments(all_k, vals, keys_s, vals_s, N, tag="m_")
        uniq2, n_uniq2 = uniq2.clone(), n_uniq2.clone()
        capm = min(N, st.total_rows)
        out_rows = eng._buf("m_rows", (capm, grad_rows.shape[1]), torch.float32)
        L.call("dmt_zero_rows", ops.p(out_rows), ops.p(n_uniq2), 0, capm, int(grad_rows.shape[1]), ops.stream_ptr())
        ws, wsb = ops.det_ws(N, int(grad_rows.shape[1]), all_r.device, "rows")
        L.call("dmt_rows_reduce_bf16" if all_r.dtype == torch.bfloat16 else "dmt_rows_reduce", ops.p(keys_s), ops.p(vals_s), ops.p(seg), N,
               st.total_rows, ops.p(all_r), ops.p(out_rows), int(grad_rows.shape[1]), ws, wsb, ops.stream_ptr())
        return (uniq2, n_uniq2, out_rows, capm)

    def train_step(self, batch: DeviceBatch, prefetch: DeviceBatch = None):
        ops.activate(self.engine.step_state)       # (stays the active one until another Trainer steps)
        try:
            return self._train_step(batch, prefetch)
        except BaseException:
            # (e.g. a collective error between backward and the optimizer: the collected weight gradients must not leak into the next step)
            ops.reset_deferred_wgrads()
            self.engine._pending_sparse = None
            self._abort_open_step()
            raise

    def _train_step(self, batch: DeviceBatch, prefetch: DeviceBatch = None):
        rank, W = parallel.world()
        dp = W > 1 or (self.force_dp and parallel.dist.is_initialized())
        plan_dp = dp and self.dp_exchange == "owner" and self.overlap_wgrads
        loss = self.forward_backward(batch, join=dp, prefetch=prefetch, defer_wgrads=plan_dp, open_step=True)
        self._inflight = None
        # (with the sparse tail of backward deferred the rows do not exist yet: reading `engine.sparse` here would hand the l2_norm row
        #  term to a stale buffer that finish_sparse_backward then zeroes)
        sparse = self.engine.sparse if self.engine._pending_sparse is None else None
        if not dp:
            lane = self._index_stream() if self.engine._pending_sparse is not None else None
            if lane is not None:
                # TWO LANES from here: the id-bound tail of the step (position / embedding-row gradients, sparse Adam: HBM-latency
                # work on few wavefronts) on the index lane, the long-row weight gradients backward skipped (MFMA work) on the compute
                # stream; they meet at the dense Adam.
                main = ops.cur_stream(self.device)
                lane.wait_stream(main)
                with torch.cuda.stream(lane):
                    self.engine.finish_sparse_backward()
                    self.opt.begin()
                    self.opt.apply_sparse(self.engine.sparse, 1.0)
                self.n_deferred = ops.run_deferred_wgrads()
                main.wait_stream(lane)
                self.opt.apply_dense(1.0)
                self.opt.end()
                self.store.refresh_shadows()
                return loss
            ops.run_deferred_wgrads()
            with self._span("optimizer"):
                self.opt.begin()
                self.opt.apply_sparse(sparse, 1.0)
                self.opt.apply_dense(1.0)
                self.opt.end()
                self.store.refresh_shadows()
            return loss
        if dp:
            early, self._early = getattr(self, "_early", None), None
            self.early_allreduce_used = early is not None
            plan = (getattr(batch, "_prep", None) or {}).get("xplan")
            if plan is not None and ops.deferred_wgrads_pending():
                # Backward ran its dX chain only (the long-row weight gradients were collected): the embedding-gradient rows are ready
                # EARLY and go on the links at once; the collected weight gradients -- MFMA work nothing else waits for -- run while
                # the rows travel (first half beside the all_to_all, second half beside the all-gather of the reduced shards).
                n_w = ops.deferred_wgrads_pending()
                t_a2a = self._mark("exchange_rows_a2a")
                handle = self.exchange_rows_begin(sparse, plan)
                with self._span("wgrads_beside_wire"):
                    ops.run_deferred_wgrads(upto=(n_w + 1) // 2)
                handle2 = self.exchange_rows_reduce(handle, plan, t_a2a)
                with self._span("wgrads_beside_wire"):
                    ops.run_deferred_wgrads()
                self.n_deferred = n_w
                t_head = self._mark("allreduce_head")
                works = [early[1] if early is not None else None,
                         parallel.allreduce_dense_(self.store.grads[: early[0]] if early is not None else self.store.grads, async_op=True, force=self.force_dp)]
                for w in works:
                    if w is not None:
                        w.wait()
                self._mark("allreduce_head", t_head)
                if early is not None:
                    self._mark("allreduce_tail", getattr(self, "_early_t0", None))
                loss = parallel.mean_scalar(loss)
                with self._span("optimizer"):
                    self.opt.begin()
                    self.opt.apply_dense(1.0 / W)
                    sparse = self.exchange_rows_collect(handle2)
                    self.opt.apply_sparse(sparse, 1.0 / W)
                    self.opt.end()
                    self.store.refresh_shadows()
                return loss
            ops.run_deferred_wgrads()          # (no exchange plan: nothing to overlap them with)
            if early is not None:
                works = [early[1], parallel.allreduce_dense_(self.store.grads[: early[0]], async_op=True, force=self.force_dp)]
            else:
                works = [parallel.allreduce_dense_(self.store.grads, async_op=True, force=self.force_dp)]
            if plan is not None:
                # no host sync between backward and the optimizer: sizes were fixed by plan_exchange before the forward pass
                t_a2a = self._mark("exchange_rows_a2a")
                handle = self.exchange_rows_begin(sparse, plan)
                t_head = self._mark("allreduce_head")
                for w in works:
                    if w is not None:
                        w.wait()
                self._mark("allreduce_head", t_head)
                if early is not None:
                    self._mark("allreduce_tail", getattr(self, "_early_t0", None))
                loss = parallel.mean_scalar(loss)
                with self._span("optimizer"):
                    self.opt.begin()
                    self.opt.apply_dense(1.0 / W)                  # overlaps the row exchange (RCCL: its own stream)
                    sparse = self.exchange_rows_collect(self.exchange_rows_reduce(handle, plan, t_a2a))
                    self.opt.apply_sparse(sparse, 1.0 / W)
                    self.opt.end()
                    self.store.refresh_shadows()
                return loss
            sparse = self.merge_sparse(sparse)
            for w in works:
                if w is not None:
                    w.wait()
            loss = parallel.mean_scalar(loss)
        self.opt.step(sparse, grad_scale=1.0 / W)
        return loss

    def close(self):
        """Release the last step's tensors (DMTEngine.release): for code that builds Trainers in a loop."""
        self.last = {}
        self._inflight = None
        self.engine.release()

    @torch.no_grad()
    def predict(self, batch: DeviceBatch):
        """run_dnn.predict scoring (run_dnn.py:663-687): sigmoid(logit + y_bias)."""
        self.sync_rows(batch, for_training=False)
        with torch.enable_grad():
            out = self.engine.inference(batch)
        (c, o), yb = out
        return torch.sigmoid(c + yb).detach(), torch.sigmoid(o + yb).detach()

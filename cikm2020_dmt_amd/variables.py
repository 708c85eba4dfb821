"""Parameter storage: flat fp32 arenas in HBM with TF-named views.

The reference creates variables lazily by scoped name under `DnnModel/`; those names are
its checkpoint surface.  Here every dense parameter lives in ONE flat fp32 arena (params / grads / Adam m, v
share the layout, so the data-parallel gradient exchange is a single RCCL all-reduce of one buffer and the
optimizer is one kernel launch), packed so that matrices the kernels want fused are contiguous:
    self/vanilla attention  dense|dense_1|dense_2 kernels -> one [d, 3d] leaf (+ [3d] bias)
    expert-{e}/expert-layer-0 weights + gates-{t} weights  -> one [K, E*512 + T*E] leaf
Embedding tables live in a second arena (p / m / v + per-row `last_step` for the exact lazy Adam).
TF-named variables are (leaf, slice) views used by state_dict()/load_state().
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import numpy as np
import torch

from . import ops
from .ops import Weight
from .spec import mmoe_input_width, trans_prefix


class LeafInfo:
    __slots__ = ("name", "shape", "offset", "numel")

    def __init__(self, name, shape, offset):
        self.name, self.shape, self.offset = name, tuple(shape), offset
        self.numel = int(np.prod(shape))


class ViewInfo:
    __slots__ = ("tf_name", "leaf", "index", "init", "shape")

    def __init__(self, tf_name, leaf, index, init, shape):
        self.tf_name, self.leaf, self.index, self.init, self.shape = tf_name, leaf, index, init, tuple(shape)


def _init_array(init: str, shape, rng) -> np.ndarray:
    """Initialisers of the reference by distribution: xavier/glorot uniform (base.py:86, tf.layers.dense),
    truncated normal sigma 0.1 (base.py:32), constants (base.py:36), zeros/ones (LayerNorm)."""
    if init == "xavier":
        lim = math.sqrt(6.0 / (shape[0] + shape[1]))
        return rng.uniform(-lim, lim, size=shape).astype(np.float32)
    if init == "trunc_normal":
        v = rng.normal(0.0, 0.1, size=shape)
        bad = np.abs(v) > 0.2
        while bad.any():
            v[bad] = rng.normal(0.0, 0.1, size=int(bad.sum()))
            bad = np.abs(v) > 0.2
        return v.astype(np.float32)
    if init.startswith("const:"):
        return np.full(shape, float(init.split(":")[1]), dtype=np.float32)
    if init == "ones":
        return np.ones(shape, dtype=np.float32)
    return np.zeros(shape, dtype=np.float32)


class VariableStore:
    def __init__(self, spec: dict, device, compute_dtype=torch.float32, seed: int = 0, init: bool = True, table_shard=None):
        """table_shard = (rank, world): ROW-SHARDED embedding tables (BASELINE configs[3]).  This process then holds, of every table,
        only the rows r with r % world == rank (p, Adam m / v and last_step alike), densely as local row r // world; every table's
        global row base is a multiple of `world`, so the owner of a GLOBAL row id g is g % world as well.  None: replicated tables."""
        self.spec = spec
        self.device = torch.device(device)
        self.compute_dtype = compute_dtype
        self.mhsa_in_use = False          # set by the engine when the fused self-attention block is selected
        self.shard = (int(table_shard[0]), int(table_shard[1])) if table_shard is not None else None
        if self.shard is not None and not (0 <= self.shard[0] < self.shard[1]):
            raise ValueError("table_shard = (rank, world) with 0 <= rank < world")
        self.leaves: Dict[str, LeafInfo] = {}
        self.views: Dict[str, ViewInfo] = {}
        self.tables: Dict[str, LeafInfo] = {}       # tf_name -> info (offset in table arena)
        self.local_rows: Dict[str, int] = {}        # tf_name -> rows held by this process (== rows when not sharded)
        self.table_rows: Dict[str, Tuple[int, int]] = {}   # tf_name -> (row_base, rows)
        self._dense_size = 0
        self._table_size = 0
        self._total_rows = 0
        self._declare_all()
        self._allocate()
        if init:
            self.initialize(seed)

    # ------------------------------------------------------------------ declaration
    def _leaf(self, name, shape):
        off = (self._dense_size + 63) // 64 * 64          # 256-byte aligned leaves
        self.leaves[name] = LeafInfo(name, shape, off)
        self._dense_size = off + int(np.prod(shape))

    def _view(self, tf_name, leaf, index, init, shape):
        self.views[tf_name] = ViewInfo(tf_name, leaf, index, init, shape)

    def _simple(self, tf_name, shape, init):
        self._leaf(tf_name, shape)
        self._view(tf_name, tf_name, (slice(None),) * len(shape), init, shape)

    def _table(self, tf_name, rows, dim):
        if tf_name in self.tables:
            return
        W = self.shard[1] if self.shard is not None else 1
        padded = (rows + W - 1) // W * W              # (sharded: every table spans a multiple of W global row ids)
        off = (self._table_size + 63) // 64 * 64
        self.tables[tf_name] = LeafInfo(tf_name, (rows, dim), off)
        self.local_rows[tf_name] = padded // W
        self._table_size = off + (padded // W) * dim
        self.table_rows[tf_name] = (self._total_rows, rows)
        self._total_rows += padded

    def _declare_all(self):
        sp = self.spec
        d, dff = sp["d_model"], sp["d_ff"]
        for (name, rows, dim, _f, _s) in sp["embedding_list"]:
            self._table("embedding_trans/%s/embedding" % name, rows, dim)
        for (name, rows, dim, _f, _s) in sp["embedding_list_bias"]:
            self._table("%s/embedding" % name, rows, dim)
        for i in range(len(sp["attention_embed_pairs"])):
            pre = trans_prefix(i)
            if sp.get("position_encoding_method", "position_learn") == "position_learn":      # (position_sin_cos adds a constant: no variable)
                self._simple(pre + "positional_encoding_k_position_learn/embedding_position_learn", (sp["maxlen_k"], d), "xavier")
            ne, nd = int(sp.get("num_blocks_encode", 1)), int(sp.get("num_blocks_decode", 1))
            for j in range(max(ne, nd)):
                blk = pre + "num_blocks_%d/" % j
                for att in (("self-attention",) if j < ne else ()) + (("vanilla_attention",) if j < nd else ()):
                    wq, bq = blk + att + "/qkv_kernel", blk + att + "/qkv_bias"
                    self._leaf(wq, (d, 3 * d))
                    self._leaf(bq, (3 * d,))
                    for c, dn in enumerate(("dense", "dense_1", "dense_2")):
                        self._view(blk + "%s/%s/kernel" % (att, dn), wq, (slice(None), slice(c * d, (c + 1) * d)), "xavier", (d, d))
                        self._view(blk + "%s/%s/bias" % (att, dn), bq, (slice(c * d, (c + 1) * d),), "zeros", (d,))
                    self._simple(blk + att + "/ln/beta", (d,), "zeros")
                    self._simple(blk + att + "/ln/gamma", (d,), "ones")
                # (encoder block j and decoder block j open the same scope 'num_blocks_j' under AUTO_REUSE: one feed-forward, TransformerModel.py:104-123,154-171)
                ffs = [blk + "positionwise_feedforward/"] if (j < ne or sp.get("tie_ffn", True)) else []
                if not sp.get("tie_ffn", True) and j < nd:
                    ffs.append(blk + "positionwise_feedforward_dec/")
                for ff in ffs:
                    self._simple(ff + "dense/kernel", (d, dff), "xavier")
                    self._simple(ff + "dense/bias", (dff,), "zeros")
                    self._simple(ff + "dense_1/kernel", (dff, d), "xavier")
                    self._simple(ff + "dense_1/bias", (d,), "zeros")
                    self._simple(ff + "ln/beta", (d,), "zeros")
                    self._simple(ff + "ln/gamma", (d,), "ones")
            if sp.get("is_trans_input_by_mlp"):
                # tf.layers.dense(seq_emb / tar_sku_emb, d_model, name='dense_trans_seq_' / 'dense_trans_sku_' + stag) (mmoe_transformer_unbias.py:196-198)
                for nm in ("seq", "sku"):
                    tp = "embedding_trans/trans_sequence_%d/dense_trans_%s_sequence_%d/" % (i, nm, i)
                    self._simple(tp + "kernel", (d, d), "xavier")
                    self._simple(tp + "bias", (d,), "zeros")
            if sp.get("is_trans_out_concat_item") and sp.get("is_trans_out_by_mlp"):
                # tf.layers.dense([user_stat, tar_sku_emb], d_model, name='dense_trans_concat_' + stag) (mmoe_transformer_unbias.py:216-217)
                tp = "embedding_trans/trans_sequence_%d/dense_trans_concat_sequence_%d/" % (i, i)
                self._simple(tp + "kernel", (2 * d, d), "xavier")
                self._simple(tp + "bias", (d,), "zeros")
        # MMoE: layer-0 of all experts and the gates share the input -> one fused [K, E*u0 + T*E] matrix
        K = mmoe_input_width(sp)
        E, T = sp["num_experts"], sp["num_tasks"]
        units = sp["hidden_units_bottom"]
        ncat = E * units[0] + T * E
        self._leaf("mmoe_layers/l0_cat_weights", (K, ncat))
        self._leaf("mmoe_layers/l0_cat_biases", (ncat,))
        for e in range(E):
            c0 = e * units[0]
            self._view("mmoe_layers/expert-%d/expert-layer-0/weights" % e, "mmoe_layers/l0_cat_weights",
                       (slice(None), slice(c0, c0 + units[0])), "trunc_normal", (K, units[0]))
            self._view("mmoe_layers/expert-%d/expert-layer-0/biases" % e, "mmoe_layers/l0_cat_biases", (slice(c0, c0 + units[0]),),
                       "const:0.1", (units[0],))
            prev = units[0]
            for li in range(1, len(units)):
                self._simple("mmoe_layers/expert-%d/expert-layer-%d/weights" % (e, li), (prev, units[li]), "trunc_normal")
                self._simple("mmoe_layers/expert-%d/expert-layer-%d/biases" % (e, li), (units[li],), "const:0.1")
                prev = units[li]
        for t in range(T):
            c0 = E * units[0] + t * E
            self._view("mmoe_layers/gates-%d/gates-layer-0/weights" % t, "mmoe_layers/l0_cat_weights",
                       (slice(None), slice(c0, c0 + E)), "trunc_normal", (K, E))
            self._view("mmoe_layers/gates-%d/gates-layer-0/biases" % t, "mmoe_layers/l0_cat_biases", (slice(c0, c0 + E),), "const:0.1", (E,))
        for name in ("click", "order")[:T]:
            prev = units[-1]
            for li, size in enumerate(sp["hidden_units_task"]):
                self._simple("%s/%s-fc-%d/weights" % (name, name, li), (prev, size), "trunc_normal")
                self._simple("%s/%s-fc-%d/biases" % (name, name, li), (size,), "const:0.1")
                prev = size
            self._simple("%s/%s-output/weights" % (name, name), (prev, 1), "trunc_normal")
            self._simple("%s/%s-output/biases" % (name, name), (1,), "const:0.1")
        prev = sum(dim for (_n, _r, dim, _f, _s) in sp["embedding_list_bias"])
        for li, size in enumerate(list(sp["hidden_units_bias"]) + [sp["output_units"]]):
            self._simple("layer_bias%d/kernel" % li, (prev, size), "xavier")
            self._simple("layer_bias%d/bias" % li, (size,), "zeros")
            prev = size

    # ------------------------------------------------------------------ allocation
    def _allocate(self):
        dev = self.device
        P = (self._dense_size + 63) // 64 * 64
        self.P = P
        self.params = torch.zeros(P, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(P, dtype=torch.float32, device=dev)
        self.adam_m = torch.zeros(P, dtype=torch.float32, device=dev)
        self.adam_v = torch.zeros(P, dtype=torch.float32, device=dev)
        TS = (self._table_size + 63) // 64 * 64
        self.tab_p = torch.zeros(TS, dtype=torch.float32, device=dev)
        self.tab_m = torch.zeros(TS, dtype=torch.float32, device=dev)
        self.tab_v = torch.zeros(TS, dtype=torch.float32, device=dev)
        W = self.shard[1] if self.shard is not None else 1
        self.last_step = torch.zeros(self._total_rows // W, dtype=torch.int32, device=dev)
        self.total_rows = self._total_rows            # size of the GLOBAL row-id space (also the "invalid key")
        self.leaf: Dict[str, torch.Tensor] = {}
        self.weight: Dict[str, Weight] = {}
        bf = self.compute_dtype == torch.bfloat16
        self.lp = torch.zeros(P, dtype=torch.bfloat16, device=dev) if bf else None
        self._w2d: List[str] = []
        # the experts' layer-li weights (li >= 1) are multiplied as ONE batched GEMM: their transposed bf16 shadows live in one
        # [E, N, Kpad] block (uniform batch stride); the fp32 / plain-bf16 leaves are already equally spaced in the arena
        batched_t: Dict[str, torch.Tensor] = {}
        if bf:
            E_, units_ = self.spec["num_experts"], self.spec["hidden_units_bottom"]
            for li in range(1, len(units_)):
                kk, nn = units_[li - 1], units_[li]
                kpad = (kk + 7) // 8 * 8
                blk = torch.zeros((E_, nn, kpad), dtype=torch.bfloat16, device=dev)
                for e in range(E_):
                    batched_t["mmoe_layers/expert-%d/expert-layer-%d/weights" % (e, li)] = blk[e][:, :kk]
        for name, info in self.leaves.items():
            t = self.params[info.offset: info.offset + info.numel].view(info.shape).detach()
            t.requires_grad_(True)
            t.grad = self.grads[info.offset: info.offset + info.numel].view(info.shape)
            self.leaf[name] = t
            if len(info.shape) == 2:
                lp = self.lp[info.offset: info.offset + info.numel].view(info.shape) if bf else None
                # transposed shadow [N, K] with the row stride padded to 8 elements (16-byte rows for vector loads, e.g. K = 3047)
                kpad = (info.shape[0] + 7) // 8 * 8
                lp_t = (batched_t[name] if name in batched_t else
                        torch.zeros((info.shape[1], kpad), dtype=torch.bfloat16, device=dev)[:, : info.shape[0]]) if bf else None
                self.weight[name] = Weight(t.detach(), lp, lp_t)
                self._w2d.append(name)
        self.table: Dict[str, torch.Tensor] = {}
        for name, info in self.tables.items():
            n_loc = self.local_rows[name] * info.shape[1]
            self.table[name] = self.tab_p[info.offset: info.offset + n_loc].view(self.local_rows[name], info.shape[1])
        # weight images of the fused feed-forward kernels (dmt_chain2), one forward + one backward image per ff scope
        self.chain: Dict[str, dict] = {}
        if bf:
            d, dff = self.spec["d_model"], self.spec["d_ff"]
            nbytes = ops.chain_image_bytes(d, dff, d)
            if nbytes is not None:
                for name in self.leaves:
                    if name.endswith("/dense/kernel") and "positionwise_feedforward" in name:
                        scope = name[: -len("dense/kernel")]
                        self.chain[scope] = dict(geo=(d, dff, d), fwd=torch.empty(nbytes, dtype=torch.uint8, device=dev),
                                                 bwd=torch.empty(nbytes, dtype=torch.uint8, device=dev))

        # streamed-weight images of the self-attention QKV projections (dmt_proj): Weight.proj, rebuilt with the shadows
        self.proj: Dict[str, torch.Tensor] = {}
        if bf:
            for name in self.leaves:
                if name.endswith("self-attention/qkv_kernel") and name in self.weight:
                    kin, n = self.leaf[name].shape
                    nbytes = ops.proj_image_bytes(int(kin), int(n))
                    if nbytes is not None:
                        self.proj[name] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                        self.weight[name].proj = self.proj[name]

        # decoder cross attention over the raw memory rows (dmt_q1mem_*): per vanilla_attention scope the V projection as a
        # k-contiguous bf16 block [d_model, d_model + 8]: row n = output column n, cols 0..d-1 = Wv[:, n], col d = bv[n], rest 0
        self.q1mem: Dict[str, torch.Tensor] = {}
        if bf and ops.q1mem_supported(self.spec["d_model"], self.spec["num_heads"], 1):
            dm = self.spec["d_model"]
            for name in self.leaves:
                if name.endswith("vanilla_attention/qkv_kernel"):
                    self.q1mem[name[: -len("qkv_kernel")]] = torch.zeros((dm, dm + 8), dtype=torch.bfloat16, device=dev)

        # weight images of the fused self-attention block (dmt_mhsa_block_fwd), one per encoder self-attention scope
        self.mhsa: Dict[str, torch.Tensor] = {}
        if bf and ops.mhsa_supported(self.spec["d_model"], self.spec["num_heads"], 1):
            nbytes = ops.mhsa_image_bytes()
            for name in self.leaves:
                if name.endswith("self-attention/qkv_kernel"):
                    self.mhsa[name[: -len("qkv_kernel")]] = torch.empty(nbytes, dtype=torch.uint8, device=dev)

        # ... and of its one-launch backward (dmt_mhsa_block_bwd: attention gradient + dx = dqkv Wqkv^T + ds), built only while an engine uses it
        self.mhsa_bwd: Dict[str, torch.Tensor] = {}
        self.mhsa_bwd_in_use = False
        if self.mhsa:
            nb = ops.mhsa_bwd_image_bytes()
            for scope in self.mhsa:
                self.mhsa_bwd[scope] = torch.empty(nb, dtype=torch.uint8, device=dev)

    # ------------------------------------------------------------------ values
    def initialize(self, seed: int = 0):
        rng = np.random.default_rng(seed)
        state = {}
        for tf_name in sorted(self.views):
            v = self.views[tf_name]
            state[tf_name] = _init_array(v.init, v.shape, rng)
        big = []
        for tf_name in sorted(self.tables):
            shape = self.tables[tf_name].shape
            if shape[0] * shape[1] > (1 << 30):
                big.append(tf_name)                   # (e.g. the 100 M-row SKU table of configs[3]: initialised on the device, below)
            else:
                state[tf_name] = _init_array("xavier", shape, rng)
        self.load_state(state)
        for i, tf_name in enumerate(big):
            rows, dim = self.tables[tf_name].shape
            lim = math.sqrt(6.0 / (rows + dim))
            g = torch.Generator(device=self.device)
            g.manual_seed(seed * 1000003 + 17 * i + (self.shard[0] if self.shard is not None else 0))
            self.table[tf_name].uniform_(-lim, lim, generator=g)

    def load_local_rows(self, tf_name: str, local_rows):
        """Row-sharded layout: this rank's rows of one table as they lie in its shard (local row l = global row l * W + rank)."""
        with torch.no_grad():
            self.table[tf_name].copy_(torch.as_tensor(np.asarray(local_rows, dtype=np.float32)).to(self.device))

    def dense_state_dict(self) -> Dict[str, np.ndarray]:
        """Every variable that is not an embedding table (name -> fp32 array).  No collective."""
        return {tf_name: self.leaf[v.leaf].detach()[v.index].float().cpu().numpy().copy() for tf_name, v in self.views.items()}

    def load_state(self, state: Dict[str, np.ndarray], refresh: bool = True):
        """state: TF variable name (Appendix B, no 'DnnModel/' prefix) -> array."""
        with torch.no_grad():
            for tf_name, arr in state.items():
                a = torch.as_tensor(np.asarray(arr, dtype=np.float32))
                if tf_name in self.views:
                    v = self.views[tf_name]
                    self.leaf[v.leaf][v.index].copy_(a.to(self.device))
                elif tf_name in self.tables:
                    if self.shard is not None:
                        r, W = self.shard
                        loc = a[r::W]
                        self.table[tf_name].zero_()
                        self.table[tf_name][: loc.shape[0]].copy_(loc.to(self.device))
                    else:
                        self.table[tf_name].copy_(a.to(self.device))
                else:
                    raise KeyError("unknown variable %s" % tf_name)
        if refresh:
            self.refresh_shadows()

    def state_dict(self) -> Dict[str, np.ndarray]:
        """Every variable as a host array.  Row-sharded layout: a COLLECTIVE that gathers every whole table onto every rank (tests and
        small tables only -- checkpoint.save / restore never call it: they move shards)."""
        out = {}
        for tf_name, v in self.views.items():
            out[tf_name] = self.leaf[v.leaf].detach()[v.index].float().cpu().numpy().copy()
        for tf_name in self.tables:
            out[tf_name] = self.full_table(tf_name).float().cpu().numpy().copy()
        return out

    def full_table(self, tf_name) -> torch.Tensor:
        """The whole [rows, dim] table.  Row-sharded layout: gathered from all ranks (a collective: every rank must call it)."""
        t = self.table[tf_name]
        if self.shard is None:
            return t
        import torch.distributed as dist
        r, W = self.shard
        rows, dim = self.tables[tf_name].shape
        if W == 1 or not (dist.is_available() and dist.is_initialized()):
            return t[:rows]
        # gathered and interleaved on the HOST (a device-side stack of W parts would hold ~2x the whole table in HBM, which is what
        # sharding is there to avoid): local row l of rank r is global row l * W + r
        if dist.get_backend() != "nccl" and t.is_cuda:
            cp = [torch.empty(t.shape, dtype=t.dtype) for _ in range(W)]
            dist.all_gather(cp, t.cpu())
        else:
            parts = [torch.empty_like(t) for _ in range(W)]
            dist.all_gather(parts, t.contiguous())
            cp = [c.cpu() for c in parts]
            del parts
        full = torch.stack(cp, dim=1).reshape(-1, dim)
        return full[:rows]

    def grad_dict(self) -> Dict[str, np.ndarray]:
        out = {}
        for tf_name, v in self.views.items():
            out[tf_name] = self.leaf[v.leaf].grad[v.index].float().cpu().numpy().copy()
        return out

    def refresh_shadows(self):
        """fp32 masters -> bf16 plain + transposed shadows (after init / load / every optimizer step)."""
        if self.compute_dtype != torch.bfloat16:
            return
        if getattr(self, "_cast_jobs", None) is None:
            triples = [(self.weight[n].f32, self.weight[n].lp, self.weight[n].lp_t) for n in self._w2d]
            dm = self.spec["d_model"]
            for scope, blk in self.q1mem.items():      # (Wv | bv) transposed into the decoder's V-projection block
                wqkv, bqkv = self.leaf[scope + "qkv_kernel"].detach(), self.leaf[scope + "qkv_bias"].detach()
                triples.append((wqkv[:, 2 * dm: 3 * dm], None, blk[:, :dm]))
                triples.append((bqkv[2 * dm: 3 * dm].view(1, dm), None, blk[:, dm: dm + 1]))
            self._cast_jobs = ops.cast_shadow_jobs(triples, self.device)
        ops.cast_shadow_batched(self._cast_jobs)
        if self.mhsa_in_use:              # (the one-launch self-attention block is optional: DMTEngine.use_mhsa)
            for scope, img in self.mhsa.items():
                ops.mhsa_image_build(self.leaf[scope + "qkv_kernel"].detach(), img)
            if self.mhsa_bwd_in_use:
                for scope, img in self.mhsa_bwd.items():
                    ops.mhsa_bwd_image_build(self.leaf[scope + "qkv_kernel"].detach(), img)
        if getattr(self, "_image_jobs", None) is None:
            jobs = ops.ImageJobs(self.device)
            for name, img in self.proj.items():
                jobs.add_proj(self.leaf[name].detach(), self.leaf[name[: -len("qkv_kernel")] + "qkv_bias"].detach(), img)
            for scope, ch in self.chain.items():
                w1, b1, w2 = self.leaf[scope + "dense/kernel"].detach(), self.leaf[scope + "dense/bias"].detach(), self.leaf[scope + "dense_1/kernel"].detach()
                # forward: A1[j, k] = W1[k, j], A2[n, j] = W2[j, n];  backward: A1[j, n] = W2[j, n], A2[k, j] = W1[k, j]
                jobs.add_chain(ch["geo"], w1, 1, w1.stride(0), w2, 1, w2.stride(0), b1, ch["fwd"])
                jobs.add_chain(ch["geo"], w2, w2.stride(0), 1, w1, w1.stride(0), 1, None, ch["bwd"])
            self._image_jobs = jobs.finish()
        self._image_jobs.run()          # every streamed-weight image (dmt_proj, dmt_chain2) in one launch

    def zero_grad(self):
        self.grads.zero_()

    def table_map(self):
        """(names, row_base[], dim[], elem_off[]) in global-row order for the sparse optimizer."""
        names = sorted(self.tables, key=lambda n: self.table_rows[n][0])
        return names, [self.table_rows[n][0] for n in names], [self.tables[n].shape[1] for n in names], [self.tables[n].offset for n in names]

    def fill_table_map(self, tm):
        """Fill a _lib.TableMap (row bases of the GLOBAL id space, element offsets into THIS process's arenas, shard)."""
        names, row_base, dims, offs = self.table_map()
        tm.n_tables = len(names)
        for i in range(len(names)):
            tm.row_base[i], tm.dim[i], tm.elem_off[i] = row_base[i], dims[i], offs[i]
        tm.row_base[len(names)] = self.total_rows
        tm.shard_r, tm.shard_w = self.shard if self.shard is not None else (0, 0)
        return tm

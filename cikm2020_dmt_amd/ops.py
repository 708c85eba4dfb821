"""Thin PyTorch plumbing over the C ABI: tensor -> (pointer, stride) marshalling and autograd Functions.

Every compute step is a libdmt_hip.so kernel; torch supplies device memory, the current HIP stream and
the autograd tape only.  Nothing here falls back to torch math: a missing library or a failed launch
raises (cikm2020_dmt_amd/_lib.py).
"""
from __future__ import annotations

import ctypes as C
import os
import ctypes as _ct
from typing import Optional

import torch

from . import _lib as L

F32, BF16 = torch.float32, torch.bfloat16

# Optional per-kernel timing with HIP events on the launch stream (bench.py's roofline leg):
#   PROFILE = {"gemm": [], "gather": []}  ->  entries (start_event, end_event, algorithmic_work)
PROFILE = None
PROFILE_KEYS = None        # None: every family is timed; a set: only these keys get events (each pair of events costs the stream ~3 us)


class _Timed:
    def __init__(self, key, work):
        self.key, self.work = key, work

    def __enter__(self):
        self.on = PROFILE is not None and (PROFILE_KEYS is None or self.key in PROFILE_KEYS)
        if self.on:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *a):
        if self.on:
            self.e1.record()
            PROFILE.setdefault(self.key, []).append((self.e0, self.e1, self.work))
        return False


# Deterministic mode (DMT_DETERMINISTIC=1 or set_deterministic(True)): the numerics mode of this ops layer, in the sense of
# torch.use_deterministic_algorithms -- every call site below then asks the library for the ordered form of a reduction (workspace
# argument / `ordered` flag), keeps GEMMs unsplit and stays away from the kernels that sum with fp32 atomics.  The LIBRARY keeps no
# mode (include/dmt_hip.h); kernel CHOICES that may differ between two engines of one process live in KernelOptions.
DETERMINISTIC = False
_det_ws = {}


def set_deterministic(on: bool):
    global DETERMINISTIC
    DETERMINISTIC = bool(on)


def det_ws(n: int, max_dim: int, device, tag=""):
    """(pointer, bytes) of the workspace dmt_embgrad_reduce / dmt_rows_reduce* need in deterministic mode; (None, 0) otherwise."""
    if not DETERMINISTIC:
        return None, 0
    need = int(L.load().dmt_reduce_det_ws_bytes(int(n), int(max_dim)))
    t = _det_ws.get((tag, str(device)))
    if t is None or t.numel() < need:
        t = torch.empty(need, dtype=torch.uint8, device=device)
        _det_ws[(tag, str(device))] = t
    return t.data_ptr(), need


_raw_stream, _cur_device = torch._C._cuda_getCurrentRawStream, torch._C._cuda_getDevice


def cur_stream(device):
    """torch.cuda.current_stream(device) without its four Python frames of device-index normalisation (12 -> 3 us; ~60 calls a step)."""
    idx = device.index
    sd = torch._C._cuda_getCurrentStream(idx if idx is not None else _cur_device())
    return torch.cuda.Stream(stream_id=sd[0], device_index=sd[1], device_type=sd[2])


def stream_ptr():
    # (the raw handle, ~0.3 us: torch.cuda.current_stream() builds a Stream object through four Python frames -- 12 us, 160 + times a step)
    return C.c_void_p(_raw_stream(_cur_device()))


def dt_code(t: torch.dtype) -> int:
    if t == F32:
        return L.DMT_F32
    if t == BF16:
        return L.DMT_BF16
    raise TypeError("unsupported dtype %s" % t)


def p(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _row_major2d(t: torch.Tensor, what: str):
    if t.dim() != 2 or (t.shape[1] > 1 and t.stride(1) != 1):
        raise ValueError("%s must be a 2-D view with unit inner stride, got shape %s strides %s" % (what, tuple(t.shape), t.stride()))
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


def require_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("libdmt_hip kernels need device tensors (got a %s tensor); there is no CPU path" % t.device)


# ------------------------------------------------------------------------------------------------ GEMM
def gemm(A, a_rs, a_cs, Bm, b_rs, b_cs, M, N, K, out, ldc, *, bias=None, act_ncols=0, gate=None, ldg=0, resid=None,
         ldr=0, ones_row=False, c_last=None, split_k=1, accumulate=False, batch=1, a_bs=0, b_bs=0, c_bs=0, bias_bs=0, gate_bs=0,
         resid_bs=0, clast_bs=0):
    require_cuda(A, Bm, out)
    d = L.GemmDesc()
    d.in_dtype = dt_code(A.dtype)
    if Bm.dtype != A.dtype:
        raise TypeError("gemm operands differ in dtype: %s vs %s" % (A.dtype, Bm.dtype))
    d.out_dtype = dt_code(out.dtype)
    d.M, d.N, d.K = M, N, K
    d.A, d.a_rs, d.a_cs = A.data_ptr(), a_rs, a_cs
    d.B, d.b_rs, d.b_cs = Bm.data_ptr(), b_rs, b_cs
    d.C, d.ldc = out.data_ptr(), ldc
    d.bias = bias.data_ptr() if bias is not None else None
    d.act_ncols = act_ncols
    d.gate, d.ldg = (gate.data_ptr(), ldg) if gate is not None else (None, 0)
    d.resid, d.ldr = (resid.data_ptr(), ldr) if resid is not None else (None, 0)
    d.a_ones_row = 1 if ones_row else 0
    d.c_last = c_last.data_ptr() if c_last is not None else None
    d.split_k, d.batch = split_k, batch
    d.accumulate = 1 if accumulate else 0
    d.a_bs, d.b_bs, d.c_bs, d.bias_bs, d.gate_bs, d.resid_bs, d.clast_bs = a_bs, b_bs, c_bs, bias_bs, gate_bs, resid_bs, clast_bs
    if PROFILE is not None:
        # algorithmic HBM bytes of this launch: every operand read once, C written once (split-K: one fp32 pass per split)
        esz, osz, nb = A.element_size(), out.element_size(), max(batch, 1)
        byt = nb * ((M * K + K * N) * esz + M * N * osz * max(split_k, 1))
        byt += nb * M * N * esz * ((gate is not None) + (resid is not None))
        PROFILE.setdefault("gemm_bytes", []).append(float(byt))
    if (_state.dw_batch is not None and accumulate and A.dtype == BF16 and out.dtype == F32 and a_cs != 1 and b_rs != 1 and K % 64 == 0
            and K // max(split_k, 1) <= 4096 and bias is None and act_ncols == 0 and gate is None and resid is None and not DETERMINISTIC):
        # a B-row weight gradient accumulated into the gradient arena: nothing reads it before the optimizer, so it waits for the others
        # of its stream and leaves with them in one launch (flush_dw_batches; dmt_gemm_dw_batched)
        st = cur_stream(A.device)
        lst = _state.dw_batch.setdefault(st, [])
        lst.append((d, (A, Bm, out, c_last), 2.0 * M * N * K * max(batch, 1)))
        if len(lst) >= DW_BATCH_MAX:
            _flush_dw_stream(st)
        return
    with _Timed("gemm_%s" % ("bf16" if A.dtype == BF16 else "f32"), 2.0 * M * N * K * max(batch, 1)):
        L.call("dmt_gemm", C.byref(d), stream_ptr())


DW_BATCH_MAX = 12          # jobs per launch (the job pack travels in the kernel argument: csrc/dmt_gemm.hip DW_MAX_JOBS)


def begin_dw_batching():
    """From now on the B-row weight-gradient GEMMs that accumulate into the gradient arena are COLLECTED per stream and launched together
    (flush_dw_batches()): a decoder's six of them are 13-17 us launches on a chip none of them fills."""
    _state.dw_batch = {}


def _flush_dw_stream(st):
    lst = _state.dw_batch.pop(st, None) if _state.dw_batch is not None else None
    if not lst:
        return 0
    n = len(lst)
    arr = (L.GemmDesc * n)(*[d for (d, _keep, _w) in lst])
    with torch.cuda.stream(st):
        with _Timed("gemm_bf16", sum(w for (_d, _k, w) in lst)):
            rc = L.load().dmt_gemm_dw_batched(arr, n, stream_ptr())
            if rc == L.DMT_ERR_UNSUPPORTED:          # (a job of another class slipped in: one by one, any class)
                for (d, _keep, _w) in lst:
                    L.call("dmt_gemm", C.byref(d), stream_ptr())
            else:
                L.check(rc, "dmt_gemm_dw_batched")
        for (_d, keep, _w) in lst:                   # (the operands may have been allocated on another stream)
            for t in keep:
                if t is not None and t.is_cuda:
                    t.record_stream(st)
    return n


def flush_dw_batches(end=False):
    """Launch what begin_dw_batching() collected, every stream's jobs on that stream; -> the streams that got a launch.  end: stop
    collecting."""
    if _state.dw_batch is None:
        return []
    out = []
    for st in list(_state.dw_batch.keys()):
        if _flush_dw_stream(st):
            out.append(st)
    if end:
        _state.dw_batch = None
    return out


def begin_ln_finish_batching():
    """From now on a LayerNorm gradient whose dgamma / dbeta go straight into the gradient arena leaves its column partials behind and
    the reduction of ALL of them is one launch (flush_ln_finish()): a train step has twelve, each a 8 us kernel between two long ones."""
    _state.ln_finish = []


def ln_bwd(dtype_code, rows, d, x, ldx, gamma, stats, dy, lddy, dx, lddx, dg, db, partials, direct):
    """dmt_ln_bwd, with the dgamma / dbeta finish either inside the call or collected for flush_ln_finish()."""
    lst = _state.ln_finish
    if lst is not None and direct:
        L.call("dmt_ln_bwd", dtype_code, rows, d, p(x), ldx, p(gamma), p(stats), p(dy), lddy, p(dx), lddx, None, None, p(partials), stream_ptr())
        lst.append((L.LnFinishJob(partials.data_ptr(), dg.data_ptr(), db.data_ptr(), int(partials.shape[0]), int(d)), (partials, dg, db),
                    cur_stream(partials.device)))
        return
    L.call("dmt_ln_bwd", dtype_code, rows, d, p(x), ldx, p(gamma), p(stats), p(dy), lddy, p(dx), lddx, p(dg), p(db), p(partials), stream_ptr())


def flush_ln_finish(end=False):
    """The collected dgamma / dbeta reductions on the CURRENT stream (which is made to wait for the streams their gradients ran on)."""
    lst = _state.ln_finish
    if lst is None:
        return 0
    _state.ln_finish = None if end else []
    if not lst:
        return 0
    cur = cur_stream(lst[0][1][0].device)
    for st in {st for (_j, _k, st) in lst}:
        if st != cur:
            cur.wait_stream(st)
    for i in range(0, len(lst), L.LN_FINISH_MAX):
        part = lst[i: i + L.LN_FINISH_MAX]
        arr = (L.LnFinishJob * len(part))(*[j for (j, _k, _s) in part])
        L.call("dmt_ln_bwd_finish_batched", arr, len(part), stream_ptr())
    for (_j, keep, _s) in lst:
        keep[0].record_stream(cur)               # (the partials were allocated on a lane)
    return len(lst)


def _pick_split(tiles: int, red: int) -> int:
    """Split-K factor of a weight-gradient GEMM.  Splitting only pays while the output tiles alone cannot fill the 256 CUs:
    every extra split adds one fp32 atomic pass over the whole output (MMoE layer-0 dW, 25 MB: split 2 is 4x slower than 1)."""
    if tiles >= 192 or DETERMINISTIC:      # (split-K partials meet in fp32 atomics)
        return 1
    s = max(1, min(1024 // max(tiles, 1), red // 512))
    return int(max(1, min(s, 512)))


class Weight:
    """A 2-D dense weight as the kernels see it: fp32 master view + (bf16 mode) plain and transposed shadows."""
    __slots__ = ("f32", "lp", "lp_t", "proj")

    def __init__(self, f32, lp=None, lp_t=None):
        self.f32, self.lp, self.lp_t = f32, lp, lp_t
        self.proj = None          # streamed-weight image of dmt_proj (VariableStore.proj), when the geometry has a kernel


def linear_forward(x, w: Weight, bias, act_ncols=0, resid=None, out=None, out_dtype=None, x_pad_finite=False):
    """y[M,N] = epi(x[M,K] @ W[K,N]).  x: 2-D row-major view (any row stride).

    x_pad_finite: the caller owns the buffer x is a view of and its columns K .. ceil8(K) hold FINITE values (not uninitialised
    memory).  With an odd K (the 3047-wide MMoE input) the bf16 reduction then runs over ceil8(K) columns -- the transposed shadow
    holds zeros there (VariableStore allocates it zeroed and only ever writes [:, :K]) -- which is the shape the direct-to-LDS
    GEMM takes (16-byte chunks)."""
    M, K = x.shape
    N = w.f32.shape[1]
    ldx = _row_major2d(x, "x")
    if x_pad_finite and x.dtype == BF16 and K % 8:
        Kp = (K + 7) // 8 * 8
        if ldx >= Kp and w.lp_t.stride(0) >= Kp and x.storage_offset() + (M - 1) * ldx + Kp <= x.untyped_storage().nbytes() // 2:
            K = Kp
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype or x.dtype, device=x.device)
    ldc = _row_major2d(out, "out")
    ldr = _row_major2d(resid, "resid") if resid is not None else 0
    if x.dtype == BF16:
        wt = w.lp_t                                  # [N, K], k contiguous
        gemm(x, ldx, 1, wt, 1, wt.stride(0), M, N, K, out, ldc, bias=bias, act_ncols=act_ncols, resid=resid, ldr=ldr)
    else:
        wf = w.f32
        gemm(x, ldx, 1, wf, wf.stride(0), 1, M, N, K, out, ldc, bias=bias, act_ncols=act_ncols, resid=resid, ldr=ldr)
    return out


def linear_backward_input(dz, w: Weight, gate=None, resid=None, out=None):
    """dx[M,K] = (dz[M,N] @ W^T) * (gate > 0) + resid."""
    M, N = dz.shape
    K = w.f32.shape[0]
    ldz = _row_major2d(dz, "dz")
    if out is None:
        # (bf16: 16-byte aligned rows also for an odd width -- the vector epilogue and the direct-to-LDS route need them)
        Kp = (K + 7) // 8 * 8 if dz.dtype == BF16 else K
        out = torch.empty((M, Kp), dtype=dz.dtype, device=dz.device)
        out = out[:, :K] if Kp != K else out
    ldc = _row_major2d(out, "out")
    wm = w.lp if dz.dtype == BF16 else w.f32          # [K, N]: B(k=n, n'=k') = W[k'*ld + n]
    gemm(dz, ldz, 1, wm, 1, wm.stride(0), M, K, N, out, ldc,
         gate=gate, ldg=_row_major2d(gate, "gate") if gate is not None else 0,
         resid=resid, ldr=_row_major2d(resid, "resid") if resid is not None else 0)
    return out


def _grad_view(leaf):
    """The fp32 gradient-arena view behind a parameter leaf (or a basic slice of one), if it can be accumulated into
    in place: 2-D with unit inner stride, or 1-D contiguous."""
    g = leaf.grad if getattr(leaf, "is_leaf", False) else None      # (.grad of a non-leaf view only warns)
    if g is None:
        base = getattr(leaf, "_base", None)
        if base is None or base.grad is None or leaf.dim() != base.dim():
            return None
        # a basic slice `base[..., a:b]` / `base[a:b]`: same strides, offset inside the base storage
        off = leaf.storage_offset() - base.storage_offset()
        if off < 0 or leaf.stride() != base.stride():
            return None
        g = base.grad.as_strided(leaf.shape, base.grad.stride(), base.grad.storage_offset() + off)
    if g.dtype != F32 or (g.dim() == 2 and g.shape[1] > 1 and g.stride(1) != 1) or (g.dim() == 1 and g.numel() > 1 and g.stride(0) != 1):
        return None
    return g


WGRAD320_MIN_ROWS = 16384      # default row threshold of the "long-row" weight gradients (a StepState may carry its own)


def _mmoe_workspace(state, Bn, dev):
    """The split expert kernels' scratch buffer (the experts' d gate partials between the backward's two launches) of this engine."""
    need = int(L.load().dmt_mmoe_experts_ws_bytes(int(Bn)))
    key = (dev.type, dev.index)
    ws = state.mmoe_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        state.mmoe_ws[key] = ws
    return ws


class StepState:
    """What ONE training step in flight keeps between its forward / backward and the point where the Trainer collects it: the long-row
    weight gradients backward only collected (deferred), the lane the MMoE / tower weight gradients fork to, the row threshold.  Every
    engine owns one (DMTEngine.step_state); Trainer.forward_backward activates its engine's for the duration of the step, so two
    Trainers with different deferral needs in one process do not see each other's (round-3 review: these were module globals).  The
    autograd engine runs backward on its own thread: the active state is a module-level pointer, not a thread-local -- steps of
    different Trainers may alternate in one process, they may not run concurrently."""
    __slots__ = ("deferred", "deferred_limit", "fork", "wgrad320_min_rows", "mmoe_ws", "unit_loss_grad", "dw_batch", "ln_finish")

    def __init__(self, wgrad320_min_rows=None):
        self.deferred = None          # list of closures while the step collects its long-row weight gradients
        self.deferred_limit = None    # only gradients that end before this element offset of the gradient arena may be collected
        self.fork = None              # dict(stream, off, n): B-row weight gradients at / behind `off` run on an idle lane
        self.wgrad320_min_rows = wgrad320_min_rows      # None: the module default above (tests lower it)
        self.mmoe_ws = {}             # device -> workspace of the split expert kernels (_mmoe_workspace)
        self.unit_loss_grad = False   # Trainer.forward_backward: the loss's incoming gradient is exactly 1 (no scaling launch)
        self.dw_batch = None          # stream -> [(GemmDesc, operands kept alive)] while B-row weight gradients are being batched
        self.ln_finish = None         # [(LnFinishJob, tensors kept alive, stream)] while the LayerNorm gradients' dgamma / dbeta sums wait

    def min_rows(self):
        return WGRAD320_MIN_ROWS if self.wgrad320_min_rows is None else self.wgrad320_min_rows


_state = StepState()        # the active one (a direct caller of the Functions, outside any Trainer, gets this default)


def activate(state):
    """Make `state` the active StepState; -> the one it replaces (hand it back to activate() when the step is over)."""
    global _state
    prev, _state = _state, (state if state is not None else StepState())
    return prev


def wgrad320_min_rows():
    return _state.min_rows()


# Weight gradients of the long (B x T)-row GEMMs can leave the backward critical path: nothing in backward reads them (only the
# optimizer does).  begin_deferred_wgrads() makes backward COLLECT them; the Trainer launches them where they hide something -- beside
# the gradient-row exchange of a data-parallel step (train_step), or beside the id-bound tail of a one-GPU step (DMT_SPARSE_LANE).
# (Tried and dropped: side streams per compute stream for them -- 1 %, and their queues collide with the lanes, streams.py.)
def begin_deferred_wgrads(limit=None):
    """From now on the long-row weight gradients of backward are COLLECTED instead of launched: nothing in backward reads them, so
    the dX chain -- the critical path to the embedding gradients -- runs through first; run_deferred_wgrads() launches them afterwards
    (Trainer.train_step: beside the id-bound tail of the step, which runs on the index lane meanwhile).
    limit: element offset into the flat gradient arena.  In a data-parallel step the arena's tail [limit:] (MMoE, towers, bias tower)
    is all-reduced from a hook DURING backward (Trainer.forward_backward): a gradient of that region must be complete when the hook
    fires, so only gradients that lie entirely before `limit` (the Transformers') are collected; the others run in place."""
    _state.deferred = []
    _state.deferred_limit = None if limit is None else int(limit)


def reset_deferred_wgrads():
    """Drop whatever a step that did not finish left collected (an exception between backward and run_deferred_wgrads)."""
    _state.deferred = None
    _state.deferred_limit = None


def _may_defer(M, *grad_views):
    if _state.deferred is None or M < _state.min_rows():
        return False
    lim = _state.deferred_limit
    if lim is None:
        return True
    for g in grad_views:
        if g is None:
            continue
        last = g.storage_offset() + sum((n - 1) * st for n, st in zip(g.shape, g.stride())) + 1 if g.numel() else g.storage_offset()
        if last > lim:
            return False
    return True


def run_deferred_wgrads(upto=None):
    """Launch what begin_deferred_wgrads() collected, on the current stream, in backward order; -> how many.  upto: only the first
    `upto` of them now (the rest stays collected for the next call)."""
    todo = _state.deferred
    if todo is None:
        return 0
    if upto is not None and upto < len(todo):
        now, _state.deferred = todo[:upto], todo[upto:]
    else:
        now, _state.deferred = todo, None
        _state.deferred_limit = None
    for fn in now:
        fn()
    return len(now)


def deferred_wgrads_pending():
    return len(_state.deferred) if _state.deferred is not None else 0


# One-GPU backward: the B-row weight gradients of the MMoE / tower region (gradient arena at or behind `from_offset`) leave the compute
# stream as they are met and run on an idle sequence lane.  That stretch of backward is ONE dependent chain of small kernels (layer-0
# input gradient <- expert kernels <- heads <- loss) with the chip mostly idle; nothing on the chain reads a weight gradient, so the
# seven GEMMs (one of them 51 GFLOP) only lengthened it.  The lanes' own work starts when dL/dz exists, i.e. after that chain.
def begin_fork_wgrads(stream, from_offset):
    _state.fork = dict(stream=stream, off=int(from_offset), n=0)


def end_fork_wgrads():
    """-> the state of begin_fork_wgrads() (n = how many launches went to the lane; the caller waits for the lane) or None."""
    st, _state.fork = _state.fork, None
    return st


def _fork_stream(M, *grad_views):
    f = _state.fork
    if f is None or M >= _state.min_rows():
        return None
    for g in grad_views:
        if g is not None and g.storage_offset() < f["off"]:
            return None
    return f["stream"]


def _deferred_wgrad320(x, dz, gw, gb, k_is_320):
    cur = cur_stream(x.device)
    x.record_stream(cur)          # (operands of the side-lane sequences were allocated on their streams)
    dz.record_stream(cur)
    if k_is_320:
        wgrad320(x, dz, gw, False, gb, 1)
    else:
        wgrad320(dz, x, gw, True, gb, 2)


def wgrad320(A, B, C, transposed, bias=None, bias_of=0):
    """C[320, N] (+)= A[M, 320]^T B[M, N]  (transposed: C[N, 320]); bias (+)= column sums of B (bias_of 1) or of A (2).  fp32 atomics."""
    d = L.WgradDesc()
    d.A, d.ld_a, d.a_cols = A.data_ptr(), A.stride(0), A.shape[1]
    d.B, d.ld_b = B.data_ptr(), B.stride(0)
    d.M, d.N = A.shape[0], B.shape[1]
    d.C, d.ldc = C.data_ptr(), C.stride(0)
    d.transposed = 1 if transposed else 0
    d.bias, d.bias_of = (bias.data_ptr(), bias_of) if bias is not None else (None, 0)
    if DETERMINISTIC:
        # ordered form: the row splits' partial blocks go through a workspace and are added in split order (no fp32 atomics).  One
        # workspace per stream: the sequence lanes run their weight gradients concurrently
        need = int(L.load().dmt_wgrad320_det_ws_bytes(int(A.shape[0]), int(B.shape[1])))
        key = ("wgrad320", str(A.device), int(cur_stream(A.device).cuda_stream))
        ws = _det_ws.get(key)
        if ws is None or ws.numel() < need:
            ws = torch.empty(need, dtype=torch.uint8, device=A.device)
            _det_ws[key] = ws
        d.det_ws, d.det_ws_bytes = ws.data_ptr(), need
    if PROFILE is not None:
        PROFILE.setdefault("wgrad320_bytes", []).append(float((A.numel() + B.numel()) * 2 + C.numel() * 4))
    with _Timed("wgrad320", 2.0 * A.shape[0] * A.shape[1] * B.shape[1]):
        L.call("dmt_wgrad320", _ct.byref(d), stream_ptr())


def _wgrad320_operand_ok(t):
    return t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % 8 == 0 and t.data_ptr() % 16 == 0


def linear_backward_weight(x, dz, want_bias=True, w_leaf=None, b_leaf=None):
    """dW[K,N] = x^T dz, db[N] = colsum(dz) (ones row), fp32, split over the (long) row dimension.
    When the parameter leaves expose their gradient-arena views the result is ACCUMULATED there by the kernel
    (fp32 atomics) and (None, None) is returned -- no temporary, no zero fill, no autograd add."""
    M, K = x.shape
    N = dz.shape[1]
    ldx, ldz = _row_major2d(x, "x"), _row_major2d(dz, "dz")
    gw = _grad_view(w_leaf) if w_leaf is not None else None
    gb = _grad_view(b_leaf) if (b_leaf is not None and want_bias) else None
    rows = K + 1 if want_bias else K
    tiles = ((rows + 127) // 128) * ((N + 127) // 128)
    split = _pick_split(tiles, M)
    if (gw is not None and (gb is not None or not want_bias) and x.dtype == BF16 and dz.dtype == BF16 and M >= _state.min_rows()
            and (K == 320 or N == 320) and K % 8 == 0 and N % 8 == 0 and _wgrad320_operand_ok(x) and _wgrad320_operand_ok(dz)
            and gw.dim() == 2 and (gw.shape[1] == 1 or gw.stride(1) == 1)):
        # the wide-block reduction kernel: the 320-wide operand is the stationary side of the [320 x 256] block
        if _may_defer(M, gw, gb if want_bias else None):
            _state.deferred.append(lambda: _deferred_wgrad320(x, dz, gw, gb if want_bias else None, K == 320))
            return None, None
        if K == 320:
            wgrad320(x, dz, gw, False, gb if want_bias else None, 1)
        else:
            wgrad320(dz, x, gw, True, gb if want_bias else None, 2)
        return None, None
    if gw is not None and (gb is not None or not want_bias):
        if _may_defer(M, gw, gb):
            def _later(x=x, dz=dz, gw=gw, gb=gb):
                cur = cur_stream(x.device)
                x.record_stream(cur)
                dz.record_stream(cur)
                gemm(x, 1, ldx, dz, ldz, 1, rows, N, M, gw, gw.stride(0) if gw.shape[0] > 1 else N, ones_row=want_bias, c_last=gb,
                     split_k=split, accumulate=True)
            _state.deferred.append(_later)
            return None, None
        side = _fork_stream(M, gw, gb)
        if side is not None:
            side.wait_stream(cur_stream(x.device))       # x, dz were produced on the current stream
            with torch.cuda.stream(side):
                x.record_stream(side)
                dz.record_stream(side)
                gemm(x, 1, ldx, dz, ldz, 1, rows, N, M, gw, gw.stride(0) if gw.shape[0] > 1 else N, ones_row=want_bias, c_last=gb,
                     split_k=split, accumulate=True)
            _state.fork["n"] += 1
            return None, None
        gemm(x, 1, ldx, dz, ldz, 1, rows, N, M, gw, gw.stride(0) if gw.shape[0] > 1 else N, ones_row=want_bias, c_last=gb,
             split_k=split, accumulate=True)
        return None, None
    dW = torch.zeros((K, N), dtype=F32, device=x.device)
    db = torch.zeros((N,), dtype=F32, device=x.device) if want_bias else None
    gemm(x, 1, ldx, dz, ldz, 1, rows, N, M, dW, N, ones_row=want_bias, c_last=db, split_k=split)
    return dW, db


def relu_bwd_(dy, y, ncols=None):
    """In place: dy[:, :ncols] *= (y[:, :ncols] > 0)."""
    rows, cols = dy.shape
    cols = cols if ncols is None else ncols
    L.call("dmt_relu_bwd", dt_code(dy.dtype), rows, cols, p(dy), _row_major2d(dy, "dy"), p(y), _row_major2d(y, "y"),
           p(dy), _row_major2d(dy, "dy"), stream_ptr())
    return dy


def relu_bwd(dy, y, ncols=None):
    """dz = dy * (y > 0) on the first ncols columns (the others pass through).  Never writes into dy (autograd owns it): all columns
    gated -> ONE out-of-place launch; a partial gate copies first."""
    rows, cols = dy.shape
    n = cols if ncols is None else ncols
    if n < cols:
        return relu_bwd_(dy.clone(), y, n)
    dz = torch.empty((rows, cols), dtype=dy.dtype, device=dy.device)
    L.call("dmt_relu_bwd", dt_code(dy.dtype), rows, cols, p(dy), _row_major2d(dy, "dy"), p(y), _row_major2d(y, "y"), p(dz), cols, stream_ptr())
    return dz


class LinearFn(torch.autograd.Function):
    """y = relu?(x W + b) (+ resid), the op behind base.dense_layer / tf.layers.dense call sites."""

    @staticmethod
    def forward(ctx, x, w_leaf, b_leaf, w: Weight, act_ncols, out_dtype, x_pad_finite=False, relu_grad_by_consumer=False):
        x2 = x.reshape(-1, x.shape[-1]) if x.dim() != 2 else x
        y = linear_forward(x2, w, b_leaf, act_ncols=act_ncols, out_dtype=out_dtype, x_pad_finite=x_pad_finite)
        ctx.pre_gated = bool(relu_grad_by_consumer)      # the consumer's backward hands over d y already times (y > 0) on the relu columns
        ctx.w = w
        ctx.leaves = (w_leaf, b_leaf)
        ctx.act_ncols = act_ncols
        ctx.xshape = x.shape
        ctx.has_bias = b_leaf is not None
        ctx.save_for_backward(x2, y if (act_ncols > 0 and not relu_grad_by_consumer) else None)
        return y.reshape(*x.shape[:-1], y.shape[-1])

    @staticmethod
    def backward(ctx, dy):
        x2, y = ctx.saved_tensors
        dz = dy.reshape(-1, dy.shape[-1])
        if dz.dtype != x2.dtype:
            dz = dz.to(x2.dtype)
        if ctx.act_ncols > 0 and not ctx.pre_gated:
            yy = y.to(dz.dtype) if y.dtype != dz.dtype else y
            dz = relu_bwd(dz, yy, ctx.act_ncols) if dz.data_ptr() == dy.data_ptr() else relu_bwd_(dz, yy, ctx.act_ncols)
        elif dz.stride(-1) != 1:
            dz = dz.contiguous()
        dx = linear_backward_input(dz, ctx.w) if ctx.needs_input_grad[0] else None
        dW, db = linear_backward_weight(x2, dz, want_bias=ctx.has_bias, w_leaf=ctx.leaves[0], b_leaf=ctx.leaves[1])
        if dx is not None:
            dx = dx.reshape(ctx.xshape)
        return dx, dW, db, None, None, None, None, None


def linear(x, w_leaf, b_leaf, w: Weight, relu=False, act_ncols=None, out_dtype=None, x_pad_finite=False, relu_grad_by_consumer=False):
    n = w.f32.shape[1]
    a = (n if relu else 0) if act_ncols is None else act_ncols
    return LinearFn.apply(x, w_leaf, b_leaf, w, a, out_dtype, x_pad_finite, relu_grad_by_consumer)


def _uniform_stride(ts):
    """Element distance between consecutive tensors of a list if it is the same for all (same dtype/shape/strides), else None."""
    if len(ts) == 1:
        return 0
    esz = ts[0].element_size()
    d0 = (ts[1].data_ptr() - ts[0].data_ptr())
    for a, b in zip(ts[:-1], ts[1:]):
        if b.data_ptr() - a.data_ptr() != d0 or a.shape != b.shape or a.stride() != b.stride() or a.dtype != b.dtype:
            return None
    return d0 // esz if d0 % esz == 0 and d0 > 0 else None


class ExpertLayerFn(torch.autograd.Function):
    """y[:, e*N:(e+1)*N] = relu(x[:, e*K:(e+1)*K] W_e + b_e) for all experts e in ONE batched GEMM (forward, input gradient and
    weight gradient each): the expert-layer-li (li >= 1) dense_layer calls of expert_gate (mmoe_transformer.py:59-79), whose
    per-expert launches (M = batch, N <= 256) are launch-latency bound."""

    @staticmethod
    def forward(ctx, x, ws, w_leaves, b_leaves):
        E = len(ws)
        M = x.shape[0]
        K, N = ws[0].f32.shape
        assert x.shape[1] == E * K
        ldx = _row_major2d(x, "x")
        y = torch.empty((M, E * N), dtype=x.dtype, device=x.device)
        Bs = [w.lp_t for w in ws] if x.dtype == BF16 else [w.f32 for w in ws]
        sb, sbias = _uniform_stride(Bs), _uniform_stride(list(b_leaves))
        if sb is not None and sbias is not None:
            if x.dtype == BF16:
                gemm(x, ldx, 1, Bs[0], 1, Bs[0].stride(0), M, N, K, y, E * N, bias=b_leaves[0], act_ncols=N, batch=E, a_bs=K, b_bs=sb, c_bs=N,
                     bias_bs=sbias)
            else:
                gemm(x, ldx, 1, Bs[0], Bs[0].stride(0), 1, M, N, K, y, E * N, bias=b_leaves[0], act_ncols=N, batch=E, a_bs=K, b_bs=sb, c_bs=N,
                     bias_bs=sbias)
        else:
            for e in range(E):
                linear_forward(x[:, e * K:(e + 1) * K], ws[e], b_leaves[e], act_ncols=N, out=y[:, e * N:(e + 1) * N])
        ctx.ws, ctx.leaves = ws, (tuple(w_leaves), tuple(b_leaves))
        ctx.save_for_backward(x, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y = ctx.saved_tensors
        ws, (w_leaves, b_leaves) = ctx.ws, ctx.leaves
        E = len(ws)
        M = x.shape[0]
        K, N = ws[0].f32.shape
        dz = dy.to(x.dtype) if dy.dtype != x.dtype else dy
        dz = relu_bwd(dz, y, E * N) if dz.data_ptr() == dy.data_ptr() else relu_bwd_(dz, y, E * N)
        ldx, ldz = _row_major2d(x, "x"), _row_major2d(dz, "dz")
        # ---- dx_e = dz_e W_e^T
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((M, E * K), dtype=dz.dtype, device=dz.device)
            Wm = [w.lp for w in ws] if dz.dtype == BF16 else [w.f32 for w in ws]
            sw = _uniform_stride(Wm)
            if sw is not None:
                gemm(dz, ldz, 1, Wm[0], 1, Wm[0].stride(0), M, K, N, dx, E * K, batch=E, a_bs=N, b_bs=sw, c_bs=K)
            else:
                for e in range(E):
                    linear_backward_input(dz[:, e * N:(e + 1) * N], ws[e], out=dx[:, e * K:(e + 1) * K])
        # ---- dW_e += x_e^T dz_e, db_e += colsum(dz_e): straight into the gradient arena
        gws = [_grad_view(l) for l in w_leaves]
        gbs = [_grad_view(l) for l in b_leaves]
        ok = all(g is not None for g in gws) and all(g is not None for g in gbs)
        sg, sgb = (_uniform_stride(gws), _uniform_stride(gbs)) if ok else (None, None)
        if sg is not None and sgb is not None:
            rows = K + 1
            tiles = E * ((rows + 127) // 128) * ((N + 127) // 128)
            gemm(x, 1, ldx, dz, ldz, 1, rows, N, M, gws[0], gws[0].stride(0), ones_row=True, c_last=gbs[0], split_k=_pick_split(tiles, M),
                 accumulate=True, batch=E, a_bs=K, b_bs=N, c_bs=sg, clast_bs=sgb)
            return dx, None, None, None
        outs = [linear_backward_weight(x[:, e * K:(e + 1) * K], dz[:, e * N:(e + 1) * N], want_bias=True, w_leaf=w_leaves[e], b_leaf=b_leaves[e])
                for e in range(E)]
        if any(o[0] is not None for o in outs):
            raise RuntimeError("ExpertLayerFn: parameter leaves without an in-place gradient view are not supported")
        return dx, None, None, None


def expert_layer(x, ws, w_leaves, b_leaves):
    return ExpertLayerFn.apply(x, list(ws), list(w_leaves), list(b_leaves))


class FFNFn(torch.autograd.Function):
    """s = relu(x W1 + b1) W2 + b2 + x  (TransformerModel_util.py:222-230 before the LayerNorm)."""

    @staticmethod
    def forward(ctx, x, w1_leaf, b1_leaf, w2_leaf, b2_leaf, w1: Weight, w2: Weight):
        x2 = x.reshape(-1, x.shape[-1])
        h = linear_forward(x2, w1, b1_leaf, act_ncols=w1.f32.shape[1])
        s = linear_forward(h, w2, b2_leaf, resid=x2)
        ctx.w1, ctx.w2 = w1, w2
        ctx.leaves = (w1_leaf, b1_leaf, w2_leaf, b2_leaf)
        ctx.save_for_backward(x2, h)
        ctx.xshape = x.shape
        return s.reshape(x.shape)

    @staticmethod
    def backward(ctx, ds):
        x2, h = ctx.saved_tensors
        ds2 = ds.reshape(-1, ds.shape[-1])
        if ds2.stride(-1) != 1:
            ds2 = ds2.contiguous()
        dh = linear_backward_input(ds2, ctx.w2, gate=h)          # (ds W2^T) * (h > 0)
        dW2, db2 = linear_backward_weight(h, ds2, w_leaf=ctx.leaves[2], b_leaf=ctx.leaves[3])
        dx = linear_backward_input(dh, ctx.w1, resid=ds2)        # dh W1^T + ds (residual branch)
        dW1, db1 = linear_backward_weight(x2, dh, w_leaf=ctx.leaves[0], b_leaf=ctx.leaves[1])
        return dx.reshape(ctx.xshape), dW1, db1, dW2, db2, None, None


# ------------------------------------------------------------------------------------------------ streamed-weight projection
def proj_image_bytes(kin, n):
    """Size of a dmt_proj weight image for out = in[., kin] W[kin, n] + b, or None when no kernel is built for the geometry."""
    if not L.load().dmt_proj_supported(kin, n):
        return None
    nb = C.c_int64(0)
    L.call("dmt_proj_image_bytes", kin, n, C.byref(nb))
    return int(nb.value)


def proj_image_build(w_f32, bias_f32, image):
    """w_f32 [kin, n] fp32 (any strides), bias [n] or None -> image (bf16, LDS byte order)."""
    L.call("dmt_proj_image_build", int(w_f32.shape[0]), int(w_f32.shape[1]), p(w_f32), w_f32.stride(0), w_f32.stride(1), p(bias_f32), p(image),
           stream_ptr())


def proj_ok(x2, w):
    img = getattr(w, "proj", None)
    return (img is not None and x2.dtype == BF16 and x2.dim() == 2 and x2.stride(1) == 1 and x2.stride(0) % 8 == 0
            and x2.data_ptr() % 16 == 0 and x2.shape[0] >= PROJ_MIN_ROWS)


def proj_forward(x2, w, n):
    """x2 [M, kin] bf16 -> [M, n] bf16 = x2 W + b by dmt_proj (the bias lives in the image: w.proj is rebuilt by the store with it)."""
    M, kin = x2.shape
    out = torch.empty((M, n), dtype=BF16, device=x2.device)
    if PROFILE is not None:
        PROFILE.setdefault("proj_bytes", []).append(float((M * kin + M * n + kin * n) * 2))
    with _Timed("proj", 2.0 * M * kin * n):
        L.call("dmt_proj", kin, n, M, p(x2), x2.stride(0), p(w.proj), p(out), n, stream_ptr())
    return out


PROJ_MIN_ROWS = 1


# ------------------------------------------------------------------------------------------------ fused ff + ln
def chain_image_bytes(kin, nmid, nout):
    """Size of a dmt_chain2 weight image for this geometry, or None when no kernel is built for it."""
    if not L.load().dmt_chain_supported(kin, nmid, nout):
        return None
    n = C.c_int64(0)
    L.call("dmt_chain_image_bytes", kin, nmid, nout, C.byref(n))
    return int(n.value)


def chain_image_build(geo, a1, a1_rs, a1_cs, a2, a2_rs, a2_cs, bias1, image):
    L.call("dmt_chain_image_build", geo[0], geo[1], geo[2], p(a1), a1_rs, a1_cs, p(a2), a2_rs, a2_cs, p(bias1), p(image), stream_ptr())


class ImageJobs:
    """Every weight image of a store rebuilt by ONE launch: add_chain / add_proj take what chain_image_build / proj_image_build
    take; finish() uploads the job table; run() is the launch.  The source tensors must stay where they are (parameter arena)."""

    def __init__(self, device):
        self.device, self.nb = device, int(L.load().dmt_image_job_bytes())
        self.host, self.dev, self.n = [], None, 0

    def _slot(self):
        buf = (C.c_uint8 * self.nb)()
        self.host.append(buf)
        return buf

    def add_chain(self, geo, a1, a1_rs, a1_cs, a2, a2_rs, a2_cs, bias1, image):
        L.call("dmt_chain_image_job", geo[0], geo[1], geo[2], p(a1), a1_rs, a1_cs, p(a2), a2_rs, a2_cs, p(bias1), p(image), C.byref(self._slot()))

    def add_proj(self, w_f32, bias_f32, image):
        L.call("dmt_proj_image_job", int(w_f32.shape[0]), int(w_f32.shape[1]), p(w_f32), w_f32.stride(0), w_f32.stride(1), p(bias_f32), p(image),
               C.byref(self._slot()))

    def finish(self):
        self.n = len(self.host)
        if self.n:
            import numpy as np
            flat = np.concatenate([np.frombuffer(b, dtype=np.uint8) for b in self.host])
            self.dev = torch.from_numpy(flat.copy()).to(self.device)
        return self

    def run(self):
        if self.n:
            L.call("dmt_image_build_batched", self.n, p(self.dev), stream_ptr())


def _chain_call(mode, geo, x2, image, M, *, bias2=None, gamma=None, beta=None, eps=0.0, s_out=None, y_out=None, stats=None, mid_out=None, mask=None):
    d = L.ChainDesc()
    d.mode, d.kin, d.nmid, d.nout, d.M = mode, geo[0], geo[1], geo[2], M
    d.in_, d.ld_in = x2.data_ptr(), x2.stride(0)
    d.image = image.data_ptr()
    d.bias2, d.gamma, d.beta, d.eps = (bias2.data_ptr() if bias2 is not None else None, gamma.data_ptr() if gamma is not None else None,
                                        beta.data_ptr() if beta is not None else None, float(eps))
    out = s_out if s_out is not None else y_out
    d.s_out = s_out.data_ptr() if s_out is not None else None
    d.y_out = y_out.data_ptr() if y_out is not None else None
    d.ld_out = out.stride(0)
    d.stats = stats.data_ptr() if stats is not None else None
    d.mid_out, d.ld_mid = (mid_out.data_ptr(), mid_out.stride(0)) if mid_out is not None else (None, 0)
    d.mask = mask.data_ptr() if mask is not None else None
    flops = 2.0 * M * geo[1] * (geo[0] + geo[2])
    with _Timed("chain2", flops):
        L.call("dmt_chain2", C.byref(d), stream_ptr())
    if PROFILE is not None:
        # algorithmic HBM bytes: input rows once, every output once, the weight image once per launch (it is re-read from L2 per tile),
        # the relu bit mask (mid / 8 bytes per row) when written
        byt = M * (geo[0] + geo[2] * ((s_out is not None) + (y_out is not None))) * 2 + (M * geo[1] * 2 if mid_out is not None else 0)
        byt += (M * geo[1] // 8) if mask is not None else 0
        PROFILE.setdefault("chain2_bytes", []).append(float(byt + image.numel()))


class FFNLNChainFn(torch.autograd.Function):
    """y = ln(relu(x W1 + b1) W2 + b2 + x): ff() + ln() of TransformerModel_util.py:212-235 as ONE launch (dmt_chain2), the
    d_ff-wide activation never re-read from memory.  Backward: LayerNorm gradient (dmt_ln_bwd), then dh / dx by the same kernel
    in its FFN_BWD mode (relu gate from the bit mask the forward wrote), weight gradients by the reduction GEMMs over h / dh."""

    @staticmethod
    def forward(ctx, x, w1_leaf, b1_leaf, w2_leaf, b2_leaf, gamma, beta, chain, eps):
        d = x.shape[-1]
        x2 = x.reshape(-1, d)
        if x2.stride(-1) != 1 or x2.stride(0) % 8 != 0 or x2.data_ptr() % 16 != 0:
            x2 = x2.contiguous()
        M = x2.shape[0]
        geo = chain["geo"]
        dev = x.device
        train = any(ctx.needs_input_grad[:7])
        y = torch.empty((M, d), dtype=BF16, device=dev)
        if train:
            s = torch.empty((M, d), dtype=BF16, device=dev)
            stats = torch.empty((M, 2), dtype=F32, device=dev)
            h = torch.empty((M, geo[1]), dtype=BF16, device=dev)
            mask = torch.empty((4 * ((M + 127) // 128), geo[1] // 32, 64), dtype=torch.int16, device=dev)
        else:
            s = stats = h = mask = None
        _chain_call(L.DMT_CHAIN_FFN_LN, geo, x2, chain["fwd"], M, bias2=b2_leaf, gamma=gamma, beta=beta, eps=eps, s_out=s, y_out=y,
                    stats=stats, mid_out=h, mask=mask)
        ctx.chain, ctx.leaves, ctx.gb = chain, (w1_leaf, b1_leaf, w2_leaf, b2_leaf), (gamma, beta)
        ctx.xshape = x.shape
        if train:
            ctx.save_for_backward(x2, s, stats, h, mask)
        return y.reshape(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, s, stats, h, mask = ctx.saved_tensors
        M, d = x2.shape
        geo = ctx.chain["geo"]
        if _state.dw_batch is not None and M >= _state.min_rows():
            # an encoder's backward begins on this lane: the decoder's B-row weight gradients collected so far leave now, in one launch,
            # in front of the long kernels (at the very end of backward they would sit beside the HBM-bound embedding tail)
            _flush_dw_stream(cur_stream(x2.device))
        dy2 = dy.reshape(-1, d)
        if dy2.stride(-1) != 1:
            dy2 = dy2.contiguous()
        if dy2.dtype != BF16:
            dy2 = dy2.to(BF16)
        gamma, beta = ctx.gb
        # ---- LayerNorm gradient: ds, dgamma / dbeta straight into the gradient arena
        ds = torch.empty((M, d), dtype=BF16, device=x2.device)
        gg, gbv = _grad_view(gamma), _grad_view(beta)
        direct = gg is not None and gbv is not None and gg.dim() == 1 and gbv.dim() == 1
        dg = gg if direct else torch.zeros((d,), dtype=F32, device=x2.device)
        db = gbv if direct else torch.zeros((d,), dtype=F32, device=x2.device)
        npart = L.load().dmt_ln_bwd_partials(M)
        partials = torch.empty((npart, 2 * d), dtype=F32, device=x2.device)
        ln_bwd(L.DMT_BF16, M, d, s, s.stride(0), gamma, stats, dy2, _row_major2d(dy2, "dy"), ds, d, dg, db, partials, direct)
        # ---- dh = (ds W2^T) * [h > 0], dx = dh W1^T + ds
        dx = torch.empty((M, d), dtype=BF16, device=x2.device)
        dh = torch.empty((M, geo[1]), dtype=BF16, device=x2.device)
        _chain_call(L.DMT_CHAIN_FFN_BWD, geo, ds, ctx.chain["bwd"], M, s_out=dx, mid_out=dh, mask=mask)
        w1_leaf, b1_leaf, w2_leaf, b2_leaf = ctx.leaves
        dW2, db2 = linear_backward_weight(h, ds, w_leaf=w2_leaf, b_leaf=b2_leaf)
        dW1, db1 = linear_backward_weight(x2, dh, w_leaf=w1_leaf, b_leaf=b1_leaf)
        return (dx.reshape(ctx.xshape), dW1, db1, dW2, db2, None if direct else dg, None if direct else db, None, None)


# ------------------------------------------------------------------------------------------------ attention
def _attn_desc(dtype, B, H, dh, Tq, Tk, q, k, v, q_lens, k_lens, resid, out, mma_fp8=False):
    d = L.AttnDesc()
    d.dtype, d.B, d.H, d.dh, d.Tq, d.Tk = dt_code(dtype), B, H, dh, Tq, Tk
    d.Q, d.q_bs, d.q_rs = q.data_ptr(), q.stride(0), q.stride(1)
    d.K, d.k_bs, d.k_rs = k.data_ptr(), k.stride(0), k.stride(1)
    d.V, d.v_bs, d.v_rs = v.data_ptr(), v.stride(0), v.stride(1)
    d.q_lens = q_lens.data_ptr() if q_lens is not None else None
    d.k_lens = k_lens.data_ptr() if k_lens is not None else None
    if resid is not None:
        d.resid, d.r_bs, d.r_rs = resid.data_ptr(), resid.stride(0), resid.stride(1)
    if out is not None:
        d.out, d.o_bs, d.o_rs = out.data_ptr(), out.stride(0), out.stride(1)
    d.mma_dtype = L.DMT_FP8_E4M3 if mma_fp8 else 0
    return d


def _chk3(t, name):
    if t.dim() != 3 or t.stride(2) != 1:
        raise ValueError("%s must be [B,T,*] with unit inner stride" % name)


def mix32(h: int) -> int:
    h &= 0xFFFFFFFF
    h ^= h >> 16; h = (h * 0x85EBCA6B) & 0xFFFFFFFF; h ^= h >> 13; h = (h * 0xC2B2AE35) & 0xFFFFFFFF; h ^= h >> 16
    return h


def site_seed(step_seed: int, stream: int) -> int:
    """Seed of one dropout site at one step (streams: 10*i+0 encoder input, +1 decoder input, +2 self-attention weights,
    +3 cross-attention weights of sequence i; 100+l bias-tower layer l)."""
    return mix32((step_seed * 0x9E3779B1 + stream * 0x85EBCA6B + 1) & 0xFFFFFFFF)


class DropoutFn(torch.autograd.Function):
    """tf.layers.dropout: y = x * keep(i) / keep_prob with libdmt_hip's counter-based mask; its own gradient."""

    @staticmethod
    def forward(ctx, x, seed32, keep_prob):
        x = x.contiguous()
        y = torch.empty_like(x)
        L.call("dmt_dropout", dt_code(x.dtype), x.numel(), p(x), p(y), int(seed32), float(keep_prob), stream_ptr())
        ctx.args = (int(seed32), float(keep_prob))
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        L.call("dmt_dropout", dt_code(dy.dtype), dy.numel(), p(dy), p(dx), ctx.args[0], ctx.args[1], stream_ptr())
        return dx, None, None


def dropout(x, rate, step_seed, stream):
    if step_seed is None or not rate:
        return x
    return DropoutFn.apply(x, site_seed(step_seed, stream), 1.0 - rate)


# ---- slicing without autograd's zero-fill + add per slice
class SplitColsFn(torch.autograd.Function):
    """x[:, a:b] column slices of one 2-D activation as ONE autograd node.  Plain slicing gives every slice its own SliceBackward
    (a zero-filled full-size buffer + a copy) and then adds the buffers; here the slices' gradients are copied side by side
    into one buffer and only the columns no slice covers are zeroed."""

    @staticmethod
    def forward(ctx, x, *bounds):
        ctx.meta = (tuple(x.shape), x.dtype, x.device, bounds)
        return tuple(x[:, bounds[2 * i]: bounds[2 * i + 1]] for i in range(len(bounds) // 2))

    @staticmethod
    def backward(ctx, *grads):
        shape, dt, dev, bounds = ctx.meta
        g = torch.empty(shape, dtype=dt, device=dev)
        spans = sorted((bounds[2 * i], bounds[2 * i + 1], i) for i in range(len(bounds) // 2))
        pos = 0
        for a, b, i in spans:
            if a < pos:
                raise RuntimeError("SplitColsFn: overlapping slices")
            if a > pos:
                g[:, pos:a].zero_()
            if grads[i] is None:
                g[:, a:b].zero_()
            else:
                g[:, a:b].copy_(grads[i])
            pos = b
        if pos < shape[1]:
            g[:, pos:].zero_()
        return (g,) + (None,) * len(bounds)


def split_cols(x, *bounds):
    return SplitColsFn.apply(x, *bounds)


class Unbind0Fn(torch.autograd.Function):
    """x[t] for every t of the leading dim as one node (backward: one stacked buffer instead of zero-fill + copy + add per t)."""

    @staticmethod
    def forward(ctx, x):
        ctx.meta = (tuple(x.shape), x.dtype, x.device)
        return tuple(x[t] for t in range(x.shape[0]))

    @staticmethod
    def backward(ctx, *grads):
        shape, dt, dev = ctx.meta
        g = torch.empty(shape, dtype=dt, device=dev)
        for t, gt in enumerate(grads):
            if gt is None:
                g[t].zero_()
            else:
                g[t].copy_(gt)
        return g


class KernelOptions:
    """Kernel choices of ONE engine (DMTEngine.kopts), handed to the autograd Functions as an argument -- no process-wide switch, so two
    Trainers in one process can differ (tests/test_gpu_boundary.py::test_two_trainers_with_different_kernel_options_in_one_process).
      attn_mma_fp8     the long-sequence (64 < T <= 256) attention forward multiplies in OCP e4m3 (Trainer(attn_dtype="fp8"))
      attn_long_fused  False: force the unfused batched-GEMM form for T > 64 (comparison runs)
      use_proj         streamed-weight QKV projection (dmt_proj) where the weight has an image; default from DMT_PROJ (1)"""
    __slots__ = ("attn_mma_fp8", "attn_long_fused", "use_proj")

    def __init__(self, attn_mma_fp8=False, attn_long_fused=True, use_proj=None):
        self.attn_mma_fp8 = bool(attn_mma_fp8)
        self.attn_long_fused = bool(attn_long_fused)
        self.use_proj = (os.environ.get("DMT_PROJ", "1") == "1") if use_proj is None else bool(use_proj)

    def replace(self, **kw):
        o = KernelOptions(self.attn_mma_fp8, self.attn_long_fused, self.use_proj)
        for k, v in kw.items():
            setattr(o, k, bool(v))
        return o


DEFAULT_OPTIONS = KernelOptions()      # what a direct caller of the Functions gets when it passes none (never mutated)


# ---- attention core: fused kernels for T <= 64 (one wavefront per (example, head)) and for 64 < T <= 256 (dmt_attn_long.hip: one
#      workgroup per (example, head), flash style); the unfused batched-GEMM form remains for what neither takes (fp32, odd head dims)
ATTN_FUSED_MAX_T = 64


def _rows16(t):
    return t is None or (t.data_ptr() % 16 == 0 and t.stride(0) % 8 == 0 and t.stride(1) % 8 == 0 and t.stride(2) == 1)


def long_fused_ok(H, *tensors, opts=None):
    """The flash-style long-sequence kernels take this call: bf16, head dim 16/32/64/80, T <= 256, 16-byte aligned rows."""
    q, k = tensors[0], tensors[1]
    if not (opts or DEFAULT_OPTIONS).attn_long_fused or q.dtype != BF16:
        return False
    if not L.load().dmt_attn_long_supported(dt_code(q.dtype), q.shape[2] // H, q.shape[1], k.shape[1]):
        return False
    return all(_rows16(t) for t in tensors)


def _long_attn_fwd(q, k, v, q_lens, k_lens, resid, out, H, drop_seed, drop_keep, causal=False):
    """T > 64 (BASELINE "long-seq variant", L = 200): S = Q K^T, P = softmax(mask(S / sqrt(dh))), out = dropout(P) V + resid as
    batched dmt_gemm launches (one per head, batch = B) around dmt_softmax_fwd.  Returns P (before dropout) for the backward."""
    B, Tq, d = q.shape
    Tk, dh = k.shape[1], d // H
    ldp = (Tk + 7) // 8 * 8
    S = torch.empty((B, H, Tq, ldp), dtype=q.dtype, device=q.device)
    P = torch.empty_like(S)
    for h in range(H):
        sl = slice(h * dh, (h + 1) * dh)
        gemm(q[..., sl], q.stride(1), 1, k[..., sl], 1, k.stride(1), Tq, Tk, dh, S[:, h], ldp, batch=B, a_bs=q.stride(0), b_bs=k.stride(0),
             c_bs=H * Tq * ldp)
    L.call("dmt_softmax_fwd", dt_code(q.dtype), B, H, Tq, Tk, p(S), ldp, p(q_lens), p(k_lens), 1.0 / float(dh) ** 0.5, int(drop_seed),
           float(drop_keep), p(P), 1 if causal else 0, stream_ptr())
    for h in range(H):
        sl = slice(h * dh, (h + 1) * dh)
        kw = {}
        if resid is not None:
            kw = dict(resid=resid[..., sl], ldr=resid.stride(1), resid_bs=resid.stride(0))
        gemm(S[:, h], ldp, 1, v[..., sl], v.stride(1), 1, Tq, dh, Tk, out[..., sl], out.stride(1), batch=B, a_bs=H * Tq * ldp, b_bs=v.stride(0),
             c_bs=out.stride(0), **kw)
    return P


def _long_attn_bwd(q, k, v, q_lens, k_lens, P, dout, dq, dk, dv, H, drop_seed, drop_keep, causal=False):
    """dP = dO V^T; dS = softmax'(P, dP); dQ = dS K; dK = dS^T Q; dV = dropout(P)^T dO -- batched GEMMs around dmt_softmax_bwd."""
    B, Tq, d = q.shape
    Tk, dh = k.shape[1], d // H
    ldp = P.shape[-1]
    dS = torch.empty_like(P)
    Pd = torch.empty_like(P)
    cb = H * Tq * ldp
    for h in range(H):
        sl = slice(h * dh, (h + 1) * dh)
        gemm(dout[..., sl], dout.stride(1), 1, v[..., sl], 1, v.stride(1), Tq, Tk, dh, dS[:, h], ldp, batch=B, a_bs=dout.stride(0),
             b_bs=v.stride(0), c_bs=cb)
    L.call("dmt_softmax_bwd", dt_code(q.dtype), B, H, Tq, Tk, p(P), p(dS), p(Pd), ldp, p(q_lens), p(k_lens), 1.0 / float(dh) ** 0.5,
           int(drop_seed), float(drop_keep), 1 if causal else 0, stream_ptr())
    for h in range(H):
        sl = slice(h * dh, (h + 1) * dh)
        gemm(dS[:, h], ldp, 1, k[..., sl], k.stride(1), 1, Tq, dh, Tk, dq[..., sl], dq.stride(1), batch=B, a_bs=cb, b_bs=k.stride(0),
             c_bs=dq.stride(0))
        gemm(dS[:, h], 1, ldp, q[..., sl], q.stride(1), 1, Tk, dh, Tq, dk[..., sl], dk.stride(1), batch=B, a_bs=cb, b_bs=q.stride(0),
             c_bs=dk.stride(0))
        gemm(Pd[:, h], 1, ldp, dout[..., sl], dout.stride(1), 1, Tk, dh, Tq, dv[..., sl], dv.stride(1), batch=B, a_bs=cb, b_bs=dout.stride(0),
             c_bs=dv.stride(0))


def attn_core_fwd(q, k, v, q_lens, k_lens, resid, out, H, drop_seed, drop_keep, opts=None, causal=False):
    """Returns what the backward needs beyond its inputs: None (fused kernels recompute P) or the saved P of the long form.
    causal (future blinding; never set by DMT's own graph) takes the unfused form at every length."""
    B, Tq, d = q.shape
    Tk = k.shape[1]
    opts = opts or DEFAULT_OPTIONS
    if causal or (max(Tq, Tk) > ATTN_FUSED_MAX_T and not long_fused_ok(H, q, k, v, resid, out, opts=opts)):
        return _long_attn_fwd(q, k, v, q_lens, k_lens, resid, out, H, drop_seed, drop_keep, causal)
    desc = _attn_desc(q.dtype, B, H, d // H, Tq, Tk, q, k, v, q_lens, k_lens, resid, out, mma_fp8=opts.attn_mma_fp8)
    desc.drop_seed, desc.drop_keep = int(drop_seed), float(drop_keep)
    with _Timed("attn_long" if max(Tq, Tk) > ATTN_FUSED_MAX_T else "attn", 4.0 * B * Tq * Tk * d):
        L.call("dmt_attn_fwd", C.byref(desc), stream_ptr())
    return None


def attn_core_bwd(q, k, v, q_lens, k_lens, P, dout, dq, dk, dv, H, drop_seed, drop_keep, causal=False, pack=None):
    B, Tq, d = q.shape
    Tk = k.shape[1]
    if P is not None:
        return _long_attn_bwd(q, k, v, q_lens, k_lens, P, dout, dq, dk, dv, H, drop_seed, drop_keep, causal)
    if pack is not None:
        # packed rows ([1, R, .] operands): one launch per length class -- the examples of at most 32 rows take the one-tile kernel
        rows = Tq
        for lo, hi, max_len in ((0, pack.n_short, 32), (pack.n_short, pack.B, 0)):
            if hi <= lo:
                continue
            bd = L.AttnBwdDesc()
            bd.f = _attn_desc(q.dtype, pack.B, H, d // H, pack.T, pack.T, q, k, v, q_lens, k_lens, None, None)
            bd.f.q_bs = bd.f.k_bs = bd.f.v_bs = 0
            bd.f.drop_seed, bd.f.drop_keep = int(drop_seed), float(drop_keep)
            bd.f.row_off, bd.f.ex_list, bd.f.n_list, bd.f.max_len = pack.row_off.data_ptr(), pack.order[lo:].data_ptr(), hi - lo, max_len
            bd.dout, bd.do_bs, bd.do_rs = dout.data_ptr(), 0, dout.stride(1)
            bd.dQ, bd.dq_bs, bd.dq_rs = dq.data_ptr(), 0, dq.stride(1)
            bd.dK, bd.dk_bs, bd.dk_rs = dk.data_ptr(), 0, dk.stride(1)
            bd.dV, bd.dv_bs, bd.dv_rs = dv.data_ptr(), 0, dv.stride(1)
            with _Timed("attn", 10.0 * rows * (hi - lo) / pack.B * (rows / pack.B) * d):
                L.call("dmt_attn_bwd", C.byref(bd), stream_ptr())
        return
    bd = L.AttnBwdDesc()
    bd.f = _attn_desc(q.dtype, B, H, d // H, Tq, Tk, q, k, v, q_lens, k_lens, None, None)
    bd.f.drop_seed, bd.f.drop_keep = int(drop_seed), float(drop_keep)
    bd.dout, bd.do_bs, bd.do_rs = dout.data_ptr(), dout.stride(0), dout.stride(1)
    bd.dQ, bd.dq_bs, bd.dq_rs = dq.data_ptr(), dq.stride(0), dq.stride(1)
    bd.dK, bd.dk_bs, bd.dk_rs = dk.data_ptr(), dk.stride(0), dk.stride(1)
    bd.dV, bd.dv_bs, bd.dv_rs = dv.data_ptr(), dv.stride(0), dv.stride(1)
    with _Timed("attn_long" if max(Tq, Tk) > ATTN_FUSED_MAX_T else "attn", 10.0 * B * Tq * Tk * d):     # S, dP, dQ, dK, dV (recomputation not counted)
        L.call("dmt_attn_bwd", C.byref(bd), stream_ptr())


class AttnFn(torch.autograd.Function):
    """out = concat_h softmax(mask(QK^T/sqrt(dh))) V + resid.  q,k,v may be column slices of packed projections;
    their gradients are written straight into one packed buffer per distinct base tensor (`pack`)."""

    @staticmethod
    def forward(ctx, packed_q, packed_kv, resid, q_lens, k_lens, H, d, self_attn, drop_seed=0, drop_keep=1.0, opts=None, causal=False):
        # self_attn: packed_q is [B,T,3d] = (Q|K|V), packed_kv is None.
        # cross:     packed_q is [B,Tq,d] = Q, packed_kv is [B,Tk,2d] = (K|V).
        if self_attn:
            _chk3(packed_q, "qkv")
            q, k, v = packed_q[..., :d], packed_q[..., d:2 * d], packed_q[..., 2 * d:]
        else:
            _chk3(packed_q, "q"); _chk3(packed_kv, "kv")
            q, k, v = packed_q, packed_kv[..., :d], packed_kv[..., d:]
        B, Tq, Tk = q.shape[0], q.shape[1], k.shape[1]
        out = torch.empty((B, Tq, d), dtype=q.dtype, device=q.device)
        ctx.P = attn_core_fwd(q, k, v, q_lens, k_lens, resid, out, H, drop_seed, drop_keep, opts, causal)
        ctx.save_for_backward(packed_q, packed_kv, q_lens, k_lens)
        ctx.H, ctx.d, ctx.self_attn, ctx.causal = H, d, self_attn, bool(causal)
        ctx.drop = (int(drop_seed), float(drop_keep))
        ctx.has_resid = resid is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        packed_q, packed_kv, q_lens, k_lens = ctx.saved_tensors
        H, d = ctx.H, ctx.d
        if dout.stride(2) != 1:
            dout = dout.contiguous()
        if ctx.self_attn:
            q, k, v = packed_q[..., :d], packed_q[..., d:2 * d], packed_q[..., 2 * d:]
            dpq = torch.empty_like(packed_q)
            dq, dk, dv = dpq[..., :d], dpq[..., d:2 * d], dpq[..., 2 * d:]
            dpkv = None
        else:
            q, k, v = packed_q, packed_kv[..., :d], packed_kv[..., d:]
            dpq = torch.empty_like(packed_q)
            dpkv = torch.empty_like(packed_kv)
            dq, dk, dv = dpq, dpkv[..., :d], dpkv[..., d:]
        attn_core_bwd(q, k, v, q_lens, k_lens, ctx.P, dout, dq, dk, dv, H, *ctx.drop, causal=ctx.causal)
        ctx.P = None
        return dpq, dpkv, (dout if ctx.has_resid else None), None, None, None, None, None, None, None, None, None


def q1mem_supported(d, H, T, dtype=BF16):
    return dtype == BF16 and bool(L.load().dmt_q1mem_supported(dt_code(dtype), int(d), int(H), int(T)))


class CrossQ1Fn(torch.autograd.Function):
    """s = multihead_attention(q_in [B, d] (one query per example), mem, mem) + q_in, WITHOUT projecting the memory to K and V
    (dmt_q1mem.hip: the projections re-associate for a single query).  q [B, d] is the already projected query (Q = q_in Wq + bq);
    w / w_leaf / b_leaf are the packed (Q|K|V) kernel and bias of the attention scope, wv_aug its [d, d+8] V block (variables.py)."""

    @staticmethod
    def forward(ctx, q, mem, q_in, k_lens, w: Weight, w_leaf, b_leaf, wv_aug, H, drop_seed, drop_keep, pack=None):
        Bn, d = q.shape
        T = mem.shape[1] if pack is None else pack.T        # (packed rows: mem is [1, R, d]; T stays the dense length)
        rows = mem.shape[0] * mem.shape[1]
        dh = d // H
        dev = q.device
        lp, ld = w.lp, w.lp.stride(0)                      # plain bf16 shadow [d, 3d]
        # q'_h = Q_h Wk[:, hc]^T  -> [B, H, d]
        qp = torch.empty((Bn, H, d), dtype=BF16, device=dev)
        gemm(q, q.stride(0), 1, lp[:, d:2 * d], 1, ld, Bn, d, dh, qp, H * d, batch=H, a_bs=dh, b_bs=dh, c_bs=d)
        cx = torch.empty((Bn, H, d + 8), dtype=BF16, device=dev)
        dd = L.Q1memDesc()
        dd.B, dd.T, dd.H, dd.d, dd.dh = Bn, T, H, d, dh
        dd.mem, dd.m_bs, dd.m_rs = mem.data_ptr(), mem.stride(0), mem.stride(1)
        if pack is not None:
            if k_lens is None or rows != pack.R:
                raise ValueError("CrossQ1Fn: packed rows need k_lens and the pack's %d rows (got %d)" % (pack.R, rows))
            dd.m_bs, dd.row_off = 0, pack.row_off.data_ptr()
        dd.k_lens = k_lens.data_ptr() if k_lens is not None else None
        dd.qp, dd.ctx, dd.ctx_hs = qp.data_ptr(), cx.data_ptr(), d + 8
        dd.drop_seed, dd.drop_keep = int(drop_seed), float(drop_keep)
        if PROFILE is not None:    # algorithmic bytes: the memory rows once, q' in, (ctx | sum P) out
            PROFILE.setdefault("q1mem_bytes", []).append(float(rows * d * 2 + Bn * H * d * 2 + Bn * H * (d + 8) * 2))
        with _Timed("q1mem", 4.0 * rows * H * d):
            L.call("dmt_q1mem_fwd", C.byref(dd), stream_ptr())
        # out[:, hc] = (ctx_h | S_h) (Wv[:, hc] ; bv_h) + q_in[:, hc]
        out = torch.empty((Bn, d), dtype=BF16, device=dev)
        gemm(cx, H * (d + 8), 1, wv_aug, 1, wv_aug.stride(0), Bn, dh, d + 8, out, d, resid=q_in, ldr=q_in.stride(0), batch=H, a_bs=d + 8,
             b_bs=dh * wv_aug.stride(0), c_bs=dh, resid_bs=dh)
        ctx.save_for_backward(q, mem, k_lens, qp, cx)
        ctx.w, ctx.leaves, ctx.dims, ctx.desc = w, (w_leaf, b_leaf), (H, d, dh, T), dd
        ctx.pack, ctx.mem_shape = pack, tuple(mem.shape)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, mem, k_lens, qp, cx = ctx.saved_tensors
        H, d, dh, T = ctx.dims
        w, (w_leaf, b_leaf) = ctx.w, ctx.leaves
        Bn = q.shape[0]
        dev = q.device
        dout = dout.contiguous()
        lp, lpt = w.lp, w.lp_t
        # d ctx_h = d out_h Wv[:, hc]^T  -> [B, H, d]
        dctx = torch.empty((Bn, H, d), dtype=BF16, device=dev)
        gemm(dout, d, 1, lp[:, 2 * d:], 1, lp.stride(0), Bn, d, dh, dctx, H * d, batch=H, a_bs=dh, b_bs=dh, c_bs=d)
        dqp = torch.empty((Bn, H, d), dtype=BF16, device=dev)
        dmem = torch.empty(ctx.mem_shape, dtype=BF16, device=dev)          # ([B, T, d], or [1, R, d] packed rows: every row is written)
        rows = ctx.mem_shape[0] * ctx.mem_shape[1]
        dd = ctx.desc
        dd.dctx, dd.dout, dd.do_bs = dctx.data_ptr(), dout.data_ptr(), d
        dd.bv = b_leaf.data_ptr() + 4 * 2 * d
        dd.dqp, dd.dmem, dd.dm_bs, dd.dm_rs = dqp.data_ptr(), dmem.data_ptr(), (0 if ctx.pack is not None else T * d), d
        if PROFILE is not None:    # algorithmic bytes: the memory rows once, d mem written once, q' / d ctx in, d q' out
            PROFILE.setdefault("q1mem_bytes", []).append(float(2 * rows * d * 2 + 3 * Bn * H * d * 2 + Bn * d * 2))
        with _Timed("q1mem", 8.0 * rows * H * d):
            L.call("dmt_q1mem_bwd", C.byref(dd), stream_ptr())
        # d Q_h = d q'_h Wk[:, hc]  -> [B, d] bf16   (B operand k-contiguous: the transposed shadow rows d + hc)
        dq = torch.empty((Bn, d), dtype=BF16, device=dev)
        ldt = lpt.stride(0)
        gemm(dqp, H * d, 1, lpt[d:2 * d], 1, ldt, Bn, dh, d, dq, d, batch=H, a_bs=d, b_bs=dh * ldt, c_bs=dh)
        # weight gradients, accumulated into the packed kernel's / bias' gradient views (reductions over the B rows only)
        gw, gb = _grad_view(w_leaf), _grad_view(b_leaf)
        if gw is None or gb is None:
            raise RuntimeError("CrossQ1Fn: parameter leaves without an in-place gradient view are not supported")
        ldg = gw.stride(0)
        split = 1 if DETERMINISTIC else max(1, min(int(os.environ.get('DMT_Q1_SPLIT', '8')), Bn // 64))
        #   dWk[:, hc] += d q'_h^T Q_h        (the K bias gets no gradient: it shifts all scores of a softmax alike)
        gemm(dqp, 1, H * d, q, q.stride(0), 1, d, dh, Bn, gw[:, d:2 * d], ldg, split_k=split, accumulate=True, batch=H, a_bs=d, b_bs=dh, c_bs=dh)
        #   dWv[:, hc] += ctx_h^T d out_h ;  dbv_h += S_h^T d out_h
        gemm(cx, 1, H * (d + 8), dout, d, 1, d, dh, Bn, gw[:, 2 * d:], ldg, split_k=split, accumulate=True, batch=H, a_bs=d + 8, b_bs=dh, c_bs=dh)
        gemm(cx[:, :, d:], 1, H * (d + 8), dout, d, 1, 1, dh, Bn, gb[2 * d:], dh, split_k=split, accumulate=True, batch=H, a_bs=d + 8, b_bs=dh,
             c_bs=dh)
        return dq, dmem, dout, None, None, None, None, None, None, None, None, None


class SelfAttnBlockFn(torch.autograd.Function):
    """s = concat_h softmax(mask(QK^T/sqrt(dh))) V + x  with (Q|K|V) = x Wqkv + b: multihead_attention(x, x, x) before its LayerNorm
    (TransformerModel_util.py:160-207) as ONE autograd node, so that dx = dqkv Wqkv^T + ds leaves the input-gradient GEMM's epilogue
    (as two nodes autograd adds the residual gradient in a separate pass over [B, T, d])."""

    @staticmethod
    def forward(ctx, x, w_leaf, b_leaf, w: Weight, lens, H, drop_seed, drop_keep, opts=None):
        _chk3(x, "x")
        B, T, d = x.shape
        x2 = x.reshape(-1, d)
        opts = opts or DEFAULT_OPTIONS
        if b_leaf is not None and opts.use_proj and proj_ok(x2, w):
            qkv = proj_forward(x2, w, 3 * d).reshape(B, T, 3 * d)        # streamed-weight projection (dmt_proj)
        else:
            qkv = linear_forward(x2, w, b_leaf).reshape(B, T, 3 * d)
        q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
        out = torch.empty((B, T, d), dtype=x.dtype, device=x.device)
        ctx.P = attn_core_fwd(q, k, v, lens, lens, x, out, H, drop_seed, drop_keep, opts)
        ctx.save_for_backward(x2, qkv, lens)
        ctx.w, ctx.leaves, ctx.H, ctx.drop = w, (w_leaf, b_leaf), H, (int(drop_seed), float(drop_keep))
        return out

    @staticmethod
    def backward(ctx, dout):
        x2, qkv, lens = ctx.saved_tensors
        B, T, d3 = qkv.shape
        d = d3 // 3
        if dout.stride(2) != 1 or dout.stride(1) != d:
            dout = dout.contiguous()
        q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
        dqkv = torch.empty_like(qkv)
        dq, dk, dv = dqkv[..., :d], dqkv[..., d:2 * d], dqkv[..., 2 * d:]
        attn_core_bwd(q, k, v, lens, lens, ctx.P, dout, dq, dk, dv, ctx.H, *ctx.drop)
        ctx.P = None
        dz = dqkv.reshape(-1, d3)
        dx = linear_backward_input(dz, ctx.w, resid=dout.reshape(-1, d)).reshape(B, T, d) if ctx.needs_input_grad[0] else None
        dW, db = linear_backward_weight(x2, dz, want_bias=ctx.leaves[1] is not None, w_leaf=ctx.leaves[0], b_leaf=ctx.leaves[1])
        return dx, dW, db, None, None, None, None, None, None


# ------------------------------------------------------------------------------------------------ fused self-attention block
def mhsa_supported(d_model, num_heads, T, B=1):
    """The one-launch block is built for the E64 geometry, T <= 64, and 32-bit byte offsets into the [B * T, 960] side output."""
    return d_model == 320 and num_heads == 4 and 1 <= T <= 64 and B * T * 1920 < 0x7FFF0000


def mhsa_image_bytes():
    n = C.c_int64(0)
    L.call("dmt_mhsa_image_bytes", C.byref(n))
    return int(n.value)


def mhsa_image_build(wqkv_f32, image):
    L.call("dmt_mhsa_image_build", p(wqkv_f32), wqkv_f32.stride(0), p(image), stream_ptr())


def mhsa_block_fwd(x, lens, image, bias, gamma, beta, eps, H, drop_seed, drop_keep, want_side=True, pack=None):
    """y, s, stats, qkv = fused multihead_attention(x, x, x) + ln (dmt_mhsa_block_fwd); side outputs None when not wanted.
    pack (engine.SeqPack): x is [1, R, d] packed rows, and so are the outputs."""
    B, T, d = x.shape
    dev = x.device
    y = torch.empty((B, T, d), dtype=BF16, device=dev)
    s = torch.empty((B, T, d), dtype=BF16, device=dev) if want_side else None      # (inference: the pre-norm sum passes through y)
    stats = torch.empty((B * T, 2), dtype=F32, device=dev) if want_side else None
    qkv = torch.empty((B, T, 3 * d), dtype=BF16, device=dev) if want_side else None
    dd = L.MhsaDesc()
    if pack is not None:
        rows = T                                    # (x.shape = (1, R, d))
        if rows != pack.R:
            raise ValueError("mhsa_block_fwd: %d packed rows given, the pack holds %d" % (rows, pack.R))
        B, T = pack.B, pack.T                       # the dense dimensions (dropout index)
        dd.blocks, dd.n_tiles, dd.n_rows = pack.blocks.data_ptr(), pack.n_tiles, pack.R
    else:
        rows = B * T
    dd.d_model, dd.num_heads, dd.B, dd.T = d, H, B, T
    dd.x, dd.lens, dd.image = x.data_ptr(), lens.data_ptr(), image.data_ptr()
    dd.bias, dd.gamma, dd.beta, dd.eps = bias.data_ptr(), gamma.data_ptr(), beta.data_ptr(), float(eps)
    dd.qkv = qkv.data_ptr() if qkv is not None else None
    dd.s_out, dd.y_out = (s.data_ptr() if s is not None else None), y.data_ptr()
    dd.stats = stats.data_ptr() if stats is not None else None
    dd.drop_seed, dd.drop_keep = int(drop_seed), float(drop_keep)
    flops = 2.0 * rows * d * 3 * d + 4.0 * rows * (T if pack is None else max(1, rows // max(B, 1))) * d
    with _Timed("mhsa_block", flops):
        L.call("dmt_mhsa_block_fwd", C.byref(dd), stream_ptr())
    if PROFILE is not None:
        PROFILE.setdefault("mhsa_block_bytes", []).append(float(rows * d * 2 * (3 + (3 if want_side else 0)) + image.numel()))
    return y, s, stats, qkv


def mhsa_bwd_image_bytes():
    n = C.c_int64(0)
    L.call("dmt_mhsa_bwd_image_bytes", C.byref(n))
    return int(n.value)


def mhsa_bwd_image_build(wqkv_f32, image):
    L.call("dmt_mhsa_bwd_image_build", p(wqkv_f32), wqkv_f32.stride(0), p(image), stream_ptr())


def mhsa_block_bwd(ds, qkv, lens, image_bwd, H, drop_seed, drop_keep, pack=None):
    """dqkv, dx = the attention gradient and dx = dqkv Wqkv^T + ds in ONE launch (dmt_mhsa_block_bwd).  ds / qkv: [B, T, d] / [B, T, 3 d]
    (pack: [1, R, .] packed rows)."""
    B, T, d = ds.shape
    dqkv = torch.empty_like(qkv)
    dx = torch.empty_like(ds)
    dd = L.MhsaBwdDesc()
    if pack is not None:
        rows = T
        B, T = pack.B, pack.T
        dd.blocks, dd.n_tiles, dd.n_rows = pack.blocks.data_ptr(), pack.n_tiles, pack.R
    else:
        rows = B * T
    dd.d_model, dd.num_heads, dd.B, dd.T = d, H, B, T
    dd.ds, dd.qkv, dd.lens, dd.image = ds.data_ptr(), qkv.data_ptr(), lens.data_ptr(), image_bwd.data_ptr()
    dd.dqkv, dd.dx = dqkv.data_ptr(), dx.data_ptr()
    dd.drop_seed, dd.drop_keep = int(drop_seed), float(drop_keep)
    # S, dP in both orientations + dQ, dK, dV (14 T^2 d / 4 ... counted as the 10 T^2 d of the unfused core: recomputation is not algorithmic work) + dx GEMM
    flops = 10.0 * rows * (T if pack is None else max(1, rows // max(B, 1))) * d + 2.0 * rows * 3 * d * d
    with _Timed("mhsa_bwd", flops):
        L.call("dmt_mhsa_block_bwd", C.byref(dd), stream_ptr())
    if PROFILE is not None:
        PROFILE.setdefault("mhsa_bwd_bytes", []).append(float(rows * d * 2 * (1 + 3 + 3 + 1) + image_bwd.numel()))
    return dqkv, dx


class MhsaBlockFn(torch.autograd.Function):
    """y = ln(x + MHA(x, x, x)): the encoder's self-attention block (TransformerModel_util.py:160-209 + ln :58-78) as ONE forward
    launch.  Backward: LayerNorm gradient, attention gradient from the saved (Q | K | V), dx = dqkv Wqkv^T + ds, weight gradient."""

    @staticmethod
    def forward(ctx, x, w_leaf, b_leaf, w: Weight, gamma, beta, lens, H, image, drop_seed, drop_keep, eps, pack=None, image_bwd=None):
        _chk3(x, "x")
        if not x.is_contiguous():
            x = x.contiguous()
        train = any(ctx.needs_input_grad[:6])
        y, s, stats, qkv = mhsa_block_fwd(x, lens, image, b_leaf, gamma, beta, eps, H, drop_seed, drop_keep, want_side=train, pack=pack)
        if train:
            ctx.save_for_backward(x, qkv, s, stats, lens)
        ctx.w, ctx.leaves, ctx.gb, ctx.H, ctx.drop = w, (w_leaf, b_leaf), (gamma, beta), H, (int(drop_seed), float(drop_keep))
        ctx.pack = pack
        ctx.image_bwd = image_bwd
        return y

    @staticmethod
    def backward(ctx, dy):
        x, qkv, s, stats, lens = ctx.saved_tensors
        B, T, d = x.shape
        M = B * T
        gamma, beta = ctx.gb
        dy2 = dy.reshape(M, d)
        if dy2.stride(-1) != 1:
            dy2 = dy2.contiguous()
        if dy2.dtype != BF16:
            dy2 = dy2.to(BF16)
        s2 = s.view(M, d)
        # ---- LayerNorm gradient (dgamma / dbeta straight into the gradient arena)
        ds = torch.empty((M, d), dtype=BF16, device=x.device)
        gg, gbv = _grad_view(gamma), _grad_view(beta)
        direct = gg is not None and gbv is not None and gg.dim() == 1 and gbv.dim() == 1
        dg = gg if direct else torch.zeros((d,), dtype=F32, device=x.device)
        db = gbv if direct else torch.zeros((d,), dtype=F32, device=x.device)
        npart = L.load().dmt_ln_bwd_partials(M)
        partials = torch.empty((npart, 2 * d), dtype=F32, device=x.device)
        ln_bwd(L.DMT_BF16, M, d, s2, d, gamma, stats, dy2, _row_major2d(dy2, "dy"), ds, d, dg, db, partials, direct)
        if ctx.image_bwd is not None:
            # ---- attention gradient + dx = dqkv Wqkv^T + ds in one launch (dmt_mhsa_block_bwd); the weight gradient reads the dqkv it wrote
            dqkv, dx = mhsa_block_bwd(ds.view(B, T, d), qkv, lens, ctx.image_bwd, ctx.H, *ctx.drop, pack=ctx.pack)
            dz = dqkv.view(M, 3 * d)
            dW, dbq = linear_backward_weight(x.view(M, d), dz, want_bias=ctx.leaves[1] is not None, w_leaf=ctx.leaves[0], b_leaf=ctx.leaves[1])
            return (dx if ctx.needs_input_grad[0] else None), dW, dbq, None, (None if direct else dg), (None if direct else db), None, None, None, None, None, None, None, None
        # ---- attention gradient
        ds3 = ds.view(B, T, d)
        q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
        dqkv = torch.empty_like(qkv)
        dq, dk, dv = dqkv[..., :d], dqkv[..., d:2 * d], dqkv[..., 2 * d:]
        attn_core_bwd(q, k, v, lens, lens, None, ds3, dq, dk, dv, ctx.H, *ctx.drop, pack=ctx.pack)
        dz = dqkv.view(M, 3 * d)
        dx = linear_backward_input(dz, ctx.w, resid=ds).view(B, T, d) if ctx.needs_input_grad[0] else None
        dW, dbq = linear_backward_weight(x.view(M, d), dz, want_bias=ctx.leaves[1] is not None, w_leaf=ctx.leaves[0], b_leaf=ctx.leaves[1])
        return dx, dW, dbq, None, (None if direct else dg), (None if direct else db), None, None, None, None, None, None, None, None


# ------------------------------------------------------------------------------------------------ LayerNorm
class LNFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        x2 = x.reshape(-1, x.shape[-1])
        if x2.stride(-1) != 1:
            x2 = x2.contiguous()
        rows, d = x2.shape
        y = torch.empty((rows, d), dtype=x.dtype, device=x.device)
        stats = torch.empty((rows, 2), dtype=F32, device=x.device)
        L.call("dmt_ln_fwd", dt_code(x.dtype), rows, d, p(x2), _row_major2d(x2, "x"), p(gamma), p(beta), float(eps), p(y), d,
               p(stats), stream_ptr())
        ctx.save_for_backward(x2, gamma, stats)
        ctx.xshape = x.shape
        ctx.gb = (gamma, beta)
        return y.reshape(x.shape)

    @staticmethod
    def backward(ctx, dy):
        x2, gamma, stats = ctx.saved_tensors
        rows, d = x2.shape
        dy2 = dy.reshape(-1, d)
        if dy2.stride(-1) != 1:
            dy2 = dy2.contiguous()
        dx = torch.empty((rows, d), dtype=x2.dtype, device=x2.device)
        # dgamma / dbeta are accumulated by the finish kernel: straight into the gradient arena when the parameters expose it
        gg, gb = _grad_view(ctx.gb[0]), _grad_view(ctx.gb[1])
        direct = gg is not None and gb is not None and gg.dim() == 1 and gb.dim() == 1
        dg = gg if direct else torch.zeros((d,), dtype=F32, device=x2.device)
        db = gb if direct else torch.zeros((d,), dtype=F32, device=x2.device)
        npart = L.load().dmt_ln_bwd_partials(rows)
        partials = torch.empty((npart, 2 * d), dtype=F32, device=x2.device)
        if dy2.dtype != x2.dtype:
            dy2 = dy2.to(x2.dtype)
        ln_bwd(dt_code(x2.dtype), rows, d, x2, _row_major2d(x2, "x"), gamma, stats, dy2, _row_major2d(dy2, "dy"), dx, d, dg, db, partials, direct)
        if direct:
            return dx.reshape(ctx.xshape), None, None, None
        return dx.reshape(ctx.xshape), dg, db, None


def layer_norm(x, gamma, beta, eps=1e-8):
    return LNFn.apply(x, gamma, beta, eps)


# ------------------------------------------------------------------------------------------------ MMoE mixture
class MixFn(torch.autograd.Function):
    """gates = softmax(glogit per task); mix[t] = sum_e gates[t][:, e] * expert[:, e*U:(e+1)*U]."""

    @staticmethod
    def forward(ctx, expert, glogit, E, U, n_tasks):
        Bn = expert.shape[0]
        gates = torch.empty((n_tasks, Bn, E), dtype=F32, device=expert.device)
        mix = torch.empty((n_tasks, Bn, U), dtype=expert.dtype, device=expert.device)
        L.call("dmt_mmoe_mix_fwd", dt_code(expert.dtype), Bn, E, U, n_tasks, p(expert), _row_major2d(expert, "expert"), p(glogit),
               _row_major2d(glogit, "glogit"), p(gates), p(mix), stream_ptr())
        ctx.save_for_backward(expert, gates)
        ctx.dims = (E, U, n_tasks)
        ctx.mark_non_differentiable(gates)
        return mix, gates

    @staticmethod
    def backward(ctx, dmix, _dgates):
        expert, gates = ctx.saved_tensors
        E, U, nt = ctx.dims
        Bn = expert.shape[0]
        dmix = dmix.contiguous()
        dexp = torch.empty((Bn, E * U), dtype=expert.dtype, device=expert.device)
        dgl = torch.empty((Bn, nt * E), dtype=expert.dtype, device=expert.device)
        L.call("dmt_mmoe_mix_bwd", dt_code(expert.dtype), Bn, E, U, nt, p(expert), _row_major2d(expert, "expert"), p(gates), p(dmix),
               p(dexp), E * U, p(dgl), nt * E, 0, stream_ptr())
        return dexp, dgl, None, None, None


def mmoe_experts_supported(units, E, T, dtype):
    return dtype == BF16 and len(units) == 3 and bool(L.load().dmt_mmoe_experts_supported(int(units[0]), int(units[1]), int(units[2]), int(E), int(T)))


class MmoeExpertsFn(torch.autograd.Function):
    """(mix [T, B, u2], gates [T, B, E]) from g1 [B, E*u0 + T*E] (the experts' relu'd layer-0 outputs | the gates' logits): expert layers 1
    and 2, the gate softmaxes and the mixtures in ONE launch (dmt_mmoe_experts_fwd); backward: one launch for the gradient of all of
    g1 (dmt_mmoe_experts_bwd) + the two batched weight-gradient GEMMs.  expert_gate, mmoe_transformer_unbias.py:63-105."""

    @staticmethod
    def forward(ctx, g1, ws1, ws2, wl1, bl1, wl2, bl2, E, T, gate_dx=False, split=True):
        Bn = g1.shape[0]
        u0, u1 = ws1[0].f32.shape
        u2 = ws2[0].f32.shape[1]
        dev = g1.device
        s1t, s2t = _uniform_stride([w.lp_t for w in ws1]), _uniform_stride([w.lp_t for w in ws2])
        s1p, s2p = _uniform_stride([w.lp for w in ws1]), _uniform_stride([w.lp for w in ws2])
        sb1, sb2 = _uniform_stride(list(bl1)), _uniform_stride(list(bl2))
        if None in (s1t, s2t, s1p, s2p, sb1, sb2):
            raise RuntimeError("MmoeExpertsFn: the experts' weights must be equally spaced in the arenas")
        d = L.MmoeDesc()
        d.B, d.E, d.T, d.u0, d.u1, d.u2 = Bn, E, T, u0, u1, u2
        d.g1, d.ldg = g1.data_ptr(), _row_major2d(g1, "g1")
        d.w1t, d.w1t_expert_stride, d.w1t_ld = ws1[0].lp_t.data_ptr(), s1t, ws1[0].lp_t.stride(0)
        d.w2t, d.w2t_expert_stride, d.w2t_ld = ws2[0].lp_t.data_ptr(), s2t, ws2[0].lp_t.stride(0)
        d.w1, d.w1_expert_stride = ws1[0].lp.data_ptr(), s1p
        d.w2, d.w2_expert_stride = ws2[0].lp.data_ptr(), s2p
        d.b1, d.b1_expert_stride, d.b2, d.b2_expert_stride = bl1[0].data_ptr(), sb1, bl2[0].data_ptr(), sb2
        h1 = torch.empty((Bn, E * u1), dtype=BF16, device=dev)
        h2 = torch.empty((Bn, E * u2), dtype=BF16, device=dev)
        gates = torch.empty((T, Bn, E), dtype=F32, device=dev)
        mix = torch.empty((T, Bn, u2), dtype=BF16, device=dev)
        d.h1, d.h2, d.gates, d.mix = h1.data_ptr(), h2.data_ptr(), gates.data_ptr(), mix.data_ptr()
        # one workgroup per (row tile, expert) -- a workspace of the engine's step state says so; gate_dx: the backward also applies the
        # relu gradient of the layer that produced g1 (ops.linear(relu_grad_by_consumer=True)): needs the split form
        ws = _mmoe_workspace(_state, Bn, dev) if (split or gate_dx) else None
        d.ws, d.ws_bytes = (ws.data_ptr(), ws.numel()) if ws is not None else (None, 0)
        d.gate_dx = 1 if gate_dx else 0
        ctx.ws = ws
        with _Timed("mmoe_experts", 2.0 * Bn * E * (u0 * u1 + u1 * u2)):
            L.call("dmt_mmoe_experts_fwd", C.byref(d), stream_ptr())
        ctx.desc = d
        ctx.leaves = (tuple(wl1), tuple(bl1), tuple(wl2), tuple(bl2))
        ctx.dims = (E, T, u0, u1, u2)
        ctx.save_for_backward(g1, h1, h2, gates)
        ctx.mark_non_differentiable(gates)
        ctx.set_materialize_grads(False)
        return mix, gates

    @staticmethod
    def backward(ctx, dmix, _dgates):
        g1, h1, h2, gates = ctx.saved_tensors
        E, T, u0, u1, u2 = ctx.dims
        wl1, bl1, wl2, bl2 = ctx.leaves
        Bn = g1.shape[0]
        dmix = dmix.contiguous()
        d = ctx.desc
        dh1, dh2 = torch.empty_like(h1), torch.empty_like(h2)
        dg1 = torch.empty((Bn, g1.shape[1]), dtype=BF16, device=g1.device)
        d.dmix, d.dh1, d.dh2, d.dg1, d.lddg = dmix.data_ptr(), dh1.data_ptr(), dh2.data_ptr(), dg1.data_ptr(), dg1.stride(0)
        with _Timed("mmoe_experts", 2.0 * Bn * E * (u0 * u1 + u1 * u2)):
            L.call("dmt_mmoe_experts_bwd", C.byref(d), stream_ptr())
        # dW_e += x_e^T dz_e, db_e += colsum(dz_e), straight into the gradient arena: one batched GEMM per layer
        for (x, ldx, K, dz, N, wls, bls) in ((g1, g1.stride(0), u0, dh1, u1, wl1, bl1), (h1, h1.stride(0), u1, dh2, u2, wl2, bl2)):
            gws, gbs = [_grad_view(l) for l in wls], [_grad_view(l) for l in bls]
            if any(g is None for g in gws) or any(g is None for g in gbs):
                raise RuntimeError("MmoeExpertsFn: parameter leaves without an in-place gradient view are not supported")
            sg, sgb = _uniform_stride(gws), _uniform_stride(gbs)
            if sg is None or sgb is None:
                raise RuntimeError("MmoeExpertsFn: the experts' gradient views must be equally spaced")
            rows = K + 1
            tiles = E * ((rows + 127) // 128) * ((N + 127) // 128)
            gemm(x, 1, ldx, dz, dz.stride(0), 1, rows, N, Bn, gws[0], gws[0].stride(0), ones_row=True, c_last=gbs[0], split_k=_pick_split(tiles, Bn),
                 accumulate=True, batch=E, a_bs=K, b_bs=N, c_bs=sg, clast_bs=sgb)
        return dg1, None, None, None, None, None, None, None, None, None, None


def heads_supported(u_in, tower_units, bias_in, bias_units, T, dtype):
    return (dtype == BF16 and not DETERMINISTIC and len(tower_units) == 1 and len(bias_units) == 2 and
            bool(L.load().dmt_heads_supported(int(u_in), int(tower_units[0]), int(bias_in), int(bias_units[0]), int(bias_units[1]), int(T))))


class HeadsFn(torch.autograd.Function):
    """(click, order, y_bias) logits [B, 1] fp32 from mix [T, B, 128] and the bias tower's input zb [B, 20]: the T task towers
    (build_tower, mmoe_transformer_unbias.py:107-126) and the position-bias tower (embedding_mlp_bias, :259-289) in ONE launch
    (dmt_heads_fwd); backward one launch (dmt_heads_bwd: input gradients + the 1-wide layers' weight gradients) + the hidden layers'
    weight-gradient GEMMs.  tower = [(fc Weight, fc w leaf, fc b leaf, out Weight, out w leaf, out b leaf)] per task; bias = three
    (Weight, w leaf, b leaf); drops = ((seed, keep), (seed, keep)) of the bias tower's two dropout sites (keep 1.0: off)."""

    @staticmethod
    def forward(ctx, mix, zb, towers, bias, drops):
        T, Bn, u_in = mix.shape
        dev = mix.device
        d = L.HeadsDesc()
        d.B, d.T, d.u_in, d.u_fc = Bn, T, u_in, towers[0][0].f32.shape[1]
        d.b_in, d.b_h0, d.b_h1 = bias[0][0].f32.shape[0], bias[0][0].f32.shape[1], bias[1][0].f32.shape[1]
        mix = mix.contiguous()
        d.mix, d.zb, d.ld_zb = mix.data_ptr(), zb.data_ptr(), _row_major2d(zb, "zb")
        for t, (fw, _fwl, fbl, ow, _owl, obl) in enumerate(towers):
            d.fc_w[t], d.fc_b[t], d.out_w[t], d.out_b[t] = fw.lp.data_ptr(), fbl.data_ptr(), ow.lp.data_ptr(), obl.data_ptr()
        for l, (w, _wl, bl) in enumerate(bias):
            d.bias_w[l], d.bias_b[l] = w.lp.data_ptr(), bl.data_ptr()
        for l in range(2):
            d.drop_seed[l], d.drop_keep[l] = int(drops[l][0]), float(drops[l][1])
        logits = torch.empty((T + 1, Bn), dtype=F32, device=dev)
        h_fc = torch.empty((T, Bn, d.u_fc), dtype=BF16, device=dev)
        h0 = torch.empty((Bn, d.b_h0), dtype=BF16, device=dev)
        h1 = torch.empty((Bn, d.b_h1), dtype=BF16, device=dev)
        d.logits, d.h_fc, d.h0, d.h1 = logits.data_ptr(), h_fc.data_ptr(), h0.data_ptr(), h1.data_ptr()
        L.call("dmt_heads_fwd", C.byref(d), stream_ptr())
        ctx.desc, ctx.towers, ctx.bias = d, towers, bias
        ctx.save_for_backward(mix, zb, h_fc, h0, h1)
        ctx.set_materialize_grads(False)
        return tuple(logits[i].unsqueeze(1) for i in range(T + 1))

    @staticmethod
    def backward(ctx, *dls):
        mix, zb, h_fc, h0, h1 = ctx.saved_tensors
        d, towers, bias = ctx.desc, ctx.towers, ctx.bias
        T, Bn, u_in = mix.shape
        dev = mix.device
        # the loss hands back three views of ONE [3, B] buffer (LossUnbiasFn.backward): use it in place when that is what arrived
        if (all(g is not None and g.dtype == F32 and g.is_contiguous() for g in dls) and
                all(dls[i].data_ptr() == dls[0].data_ptr() + 4 * Bn * i for i in range(T + 1))):
            dl = dls[0]
        else:
            dl = torch.empty((T + 1, Bn), dtype=F32, device=dev)
            for i, g in enumerate(dls):
                if g is None:
                    dl[i].zero_()
                else:
                    dl[i].copy_(g.reshape(-1))
        dmix = torch.empty_like(mix)
        dzb = torch.empty((Bn, d.b_in), dtype=BF16, device=dev)
        dz_fc, dz0, dz1 = torch.empty_like(h_fc), torch.empty_like(h0), torch.empty_like(h1)
        d.dlogits, d.dmix, d.dzb, d.ld_dzb = dl.data_ptr(), dmix.data_ptr(), dzb.data_ptr(), dzb.stride(0)
        d.dz_fc, d.dz0, d.dz1 = dz_fc.data_ptr(), dz0.data_ptr(), dz1.data_ptr()
        for t, (_fw, _fwl, _fbl, _ow, owl, obl) in enumerate(towers):
            gw, gb = _grad_view(owl), _grad_view(obl)
            if gw is None or gb is None:
                raise RuntimeError("HeadsFn: parameter leaves without an in-place gradient view are not supported")
            d.g_out_w[t], d.g_out_b[t] = gw.data_ptr(), gb.data_ptr()
        gw2, gb2 = _grad_view(bias[2][1]), _grad_view(bias[2][2])
        if gw2 is None or gb2 is None:
            raise RuntimeError("HeadsFn: parameter leaves without an in-place gradient view are not supported")
        d.g_bias_w2, d.g_bias_b2 = gw2.data_ptr(), gb2.data_ptr()
        L.call("dmt_heads_bwd", C.byref(d), stream_ptr())
        # hidden layers: dW += x^T dz, db += colsum(dz), accumulated into the gradient arena
        for t, (_fw, fwl, fbl, _ow, _owl, _obl) in enumerate(towers):
            linear_backward_weight(mix[t], dz_fc[t], want_bias=True, w_leaf=fwl, b_leaf=fbl)
        linear_backward_weight(zb, dz0, want_bias=True, w_leaf=bias[0][1], b_leaf=bias[0][2])
        linear_backward_weight(h0, dz1, want_bias=True, w_leaf=bias[1][1], b_leaf=bias[1][2])
        return dmix, dzb, None, None, None


# ------------------------------------------------------------------------------------------------ loss
class LossUnbiasFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, click, order, ybias, mask5, w_ctr, w_ecvr, lw, method, ctr_rel):
        Bn = click.numel()
        c, o, yb = (t.reshape(-1).to(F32).contiguous() for t in (click, order, ybias))
        loss = torch.empty((1,), dtype=F32, device=c.device)
        pc = torch.empty((Bn,), dtype=F32, device=c.device)
        pv = torch.empty((Bn,), dtype=F32, device=c.device)
        dall = torch.empty((3, Bn), dtype=F32, device=c.device)       # one buffer: the backward scales all three with one launch
        dc, do, db = dall[0], dall[1], dall[2]
        L.call("dmt_loss_unbias", Bn, p(c), p(o), p(yb), p(mask5), p(w_ctr), p(w_ecvr), float(lw[0]), float(lw[1]), int(method),
               int(ctr_rel), 1.0, p(loss), p(pc), p(pv), p(dc), p(do), p(db), stream_ptr())
        ctx.save_for_backward(dall)
        ctx.shapes = (click.shape, order.shape, ybias.shape, click.dtype, order.dtype, ybias.dtype)
        ctx.mark_non_differentiable(pc, pv)
        ctx.set_materialize_grads(False)
        return loss.reshape(()), pc, pv

    @staticmethod
    def backward(ctx, gloss, _a, _b):
        (dall,) = ctx.saved_tensors
        s0, s1, s2, t0, t1, t2 = ctx.shapes
        if gloss is None:
            return (None,) * 9
        g = dall if _state.unit_loss_grad else dall * gloss
        return (g[0].reshape(s0).to(t0), g[1].reshape(s1).to(t1), g[2].reshape(s2).to(t2), None, None, None, None, None, None)


class ScaleAddPosFn(torch.autograd.Function):
    """y = scale * x + pos[:T]  (TransformerModel.py:96-100)."""

    @staticmethod
    def forward(ctx, x, pos, scale):
        x = x.contiguous()
        Bn, T, d = x.shape
        y = torch.empty_like(x)
        L.call("dmt_scale_add_pos", dt_code(x.dtype), Bn, T, d, p(x), float(scale), p(pos), p(y), stream_ptr())
        ctx.scale, ctx.shape = scale, (Bn, T, d)
        ctx.pos_shape = tuple(pos.shape) if pos is not None else None
        return y

    @staticmethod
    def backward(ctx, dy):
        Bn, T, d = ctx.shape
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        L.call("dmt_scale_add_pos", dt_code(dy.dtype), Bn, T, d, p(dy), float(ctx.scale), None, p(dx), stream_ptr())
        dpos = None
        if ctx.pos_shape is not None:
            dpos = torch.zeros(ctx.pos_shape, dtype=F32, device=dy.device)
            colsum(dy.view(Bn, T * d), 1.0, out=dpos.view(-1)[: T * d])
        return dx, dpos, None


# ------------------------------------------------------------------------------------------------ misc
def cast_shadow(src_f32_2d, dst_plain, dst_t):
    rows, cols = src_f32_2d.shape
    L.call("dmt_cast_transpose_bf16", rows, cols, p(src_f32_2d), src_f32_2d.stride(0), p(dst_plain),
           dst_plain.stride(0) if dst_plain is not None else 0, p(dst_t), dst_t.stride(0) if dst_t is not None else 0, stream_ptr())


def cast_shadow_jobs(triples, device):
    """Device job table for cast_shadow_batched: triples of (fp32 master [K, N], bf16 plain or None, bf16 transposed or None)."""
    import ctypes as C
    import numpy as np
    from ._lib import CastJob
    jobs = (CastJob * len(triples))()
    tiles = 0
    for j, (src, dp, dt) in enumerate(triples):
        rows, cols = src.shape
        tx, ty = (cols + 31) // 32, (rows + 31) // 32
        jobs[j] = CastJob(p(src), p(dp), p(dt), src.stride(0), dp.stride(0) if dp is not None else 0,
                          dt.stride(0) if dt is not None else 0, rows, cols, tiles, tx)
        tiles += tx * ty
    raw = np.frombuffer(C.string_at(C.addressof(jobs), C.sizeof(jobs)), dtype=np.uint8).copy()
    return torch.from_numpy(raw).to(device), len(triples), tiles


def cast_shadow_batched(table):
    dev_jobs, n, tiles = table
    L.call("dmt_cast_transpose_bf16_batched", n, p(dev_jobs), tiles, stream_ptr())


def colsum(x2d, scale=1.0, out=None):
    rows, cols = x2d.shape
    if out is None:
        out = torch.zeros((cols,), dtype=F32, device=x2d.device)
    L.call("dmt_colsum", dt_code(x2d.dtype), rows, cols, p(x2d), _row_major2d(x2d, "x"), float(scale), p(out), 1 if DETERMINISTIC else 0, stream_ptr())
    return out

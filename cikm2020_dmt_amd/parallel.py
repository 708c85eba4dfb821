"""Data-parallel exchange: one process per GPU, torch.distributed (backend 'nccl' == RCCL over xGMI).

Reference semantics (run_dnn.py:45-87,148-207): every tower computes the loss mean over ITS batch, gradients are
averaged over towers (mean of means) and one Adam step is applied.  Here:
  * dense parameters: ONE all-reduce(sum) of the flat fp32 gradient arena (13.7 MB at reference dims), scaled by
    1/world inside the Adam kernel;
  * embedding tables: never densified.  Each rank's (row id, fp32 grad row) pairs -- already reduced per row on the
    GPU -- are all-gathered (padded to the largest rank), concatenated in RANK ORDER and reduced again per row by
    the same stable-sort + segment-reduce kernels, so every rank applies the identical update.
The functions below only move data; they work on CPU tensors with the gloo backend (tests) and on device tensors
with RCCL.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


def world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def allreduce_dense_(flat_grads: torch.Tensor, async_op: bool = False):
    """Sum the flat gradient arena over ranks in place (the mean is applied by the optimizer's grad_scale)."""
    if world()[1] == 1:
        return None
    return dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, async_op=async_op)


def _all_gather_cat(dst: torch.Tensor, loc: torch.Tensor, W: int, cap: int):
    if dist.get_backend() == "nccl" and hasattr(dist, "all_gather_into_tensor"):
        dist.all_gather_into_tensor(dst, loc)      # one RCCL all-gather straight into the rank-major buffer
        return
    parts = [torch.empty_like(loc) for _ in range(W)]
    dist.all_gather(parts, loc)
    for r in range(W):
        dst[r * cap:(r + 1) * cap] = parts[r]


def allgather_sparse(keys: torch.Tensor, rows: torch.Tensor, n: int, invalid_key: int, transport_dtype=None):
    """keys [>=n] int32 (global row ids), rows [>=n, D] fp32, n valid entries on this rank.
    Returns (all_keys [W*cap], all_rows [W*cap, D], cap): rank-major concatenation; unused slots carry
    `invalid_key` (they sort last and are skipped by the reduce kernels; their rows are never read).
    `transport_dtype` (torch.bfloat16 in bf16 mode) is the wire format of the rows: per-rank row sums are rounded once
    before the cross-rank sum, which halves the 8-rank exchange (~147 MB of fp32 rows per rank and step at E64)."""
    rank, W = world()
    dev = keys.device
    if W > 1:
        cnt = torch.tensor([n], dtype=torch.int64, device=dev)
        cnts = [torch.zeros_like(cnt) for _ in range(W)]
        dist.all_gather(cnts, cnt)
        cap = int(max(int(c.item()) for c in cnts))
    else:
        cap = n
    cap = max(cap, 1)
    D = rows.shape[1]
    if keys.shape[0] >= cap and rows.shape[0] >= cap and rows.is_contiguous():
        k_loc = keys[:cap].clone()                 # no zero-filled staging copy of the rows: slots past n are never read
        k_loc[n:] = invalid_key
        r_loc = rows[:cap]
    else:
        k_loc = torch.full((cap,), invalid_key, dtype=keys.dtype, device=dev)
        r_loc = torch.zeros((cap, D), dtype=rows.dtype, device=dev)
        k_loc[:n] = keys[:n]
        r_loc[:n] = rows[:n]
    if transport_dtype is not None and transport_dtype != r_loc.dtype:
        r_loc = r_loc.to(transport_dtype)
    if W == 1:
        return k_loc, r_loc, cap
    all_k = torch.empty((W * cap,), dtype=keys.dtype, device=dev)
    all_r = torch.empty((W * cap, D), dtype=r_loc.dtype, device=dev)
    _all_gather_cat(all_k, k_loc, W, cap)
    _all_gather_cat(all_r, r_loc.contiguous(), W, cap)
    return all_k, all_r, cap


def mean_scalar(x: torch.Tensor) -> torch.Tensor:
    """average_losses (run_dnn.py:83-87): mean over towers of the per-tower mean loss."""
    rank, W = world()
    if W == 1:
        return x
    y = x.detach().clone()
    dist.all_reduce(y, op=dist.ReduceOp.SUM)
    return y / W

"""Data-parallel exchange: one process per GPU, torch.distributed (backend 'nccl' == RCCL over xGMI).

Reference semantics (run_dnn.py:45-87,148-207): every tower computes the loss mean over ITS batch, gradients are
averaged over towers (mean of means) and one Adam step is applied.  Here:
  * dense parameters: ONE all-reduce(sum) of the flat fp32 gradient arena (13.7 MB at reference dims), scaled by
    1/world inside the Adam kernel;
  * embedding tables: never densified.  Each rank's (row id, grad row) pairs -- already reduced per row on the GPU -- go to
    the row's OWNER rank (row % world) by all_to_all_single, are concatenated there in RANK ORDER and reduced again per row by
    the same stable-sort + segment-reduce kernels; the reduced shards are all-gathered, so every rank applies the identical
    update to its replica (exchange_to_owners / allgather_shards).  The older one-step form (all-gather everything, every
    rank reduces everything: allgather_sparse) is kept behind Trainer(dp_exchange="allgather").
The functions below only move data; they work on CPU tensors with the gloo backend (tests) and on device tensors
with RCCL.
"""
from __future__ import annotations

import os
from typing import Tuple

import torch
import torch.distributed as dist


def world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


_index_group = {"world": None, "group": None}


def index_group():
    """The communicator of the INDEX PLANE (row ids, counts: Trainer.plan_exchange).  RCCL: a second communicator, hence its own
    stream -- the id exchange of batch i + 1 is not ordered behind the gradient collectives of batch i, so it can run (and its two
    host syncs can return) while step i computes.  Created collectively on first use (every rank reaches its first plan_exchange at
    the same point of the program); other backends share the default group.
    Hazard (round-2 advice): two communicators issue their kernels from two streams, and NCCL / RCCL only guarantee progress when the
    collectives of different communicators are launched in the same order on every rank or can co-reside on the device.  Here each
    rank issues, per step, the index plane of batch i + 1 (this communicator) and the gradient collectives of step i (the default one)
    from different host points, so the device-side order can differ between ranks; both kinds of kernels are small (a few workgroups) and
    co-reside on a 256-CU device, which is what makes this work -- it has not been soak-tested on 8 GPUs.
    DEFAULT SINCE ROUND 6: DMT_INDEX_GROUP=0 -- everything on the default communicator (one stream, one order on every rank: the id
    exchange queues behind the gradient collectives of the step before; progress then rests on nothing but NCCL's own in-order rule).
    DMT_INDEX_GROUP=1 opts into the second communicator once an 8-rank soak on hardware has shown it safe."""
    mode = os.environ.get("DMT_INDEX_GROUP", "0")
    if not (dist.is_available() and dist.is_initialized()) or mode not in ("1", "force") or (dist.get_backend() != "nccl" and mode != "force"):
        return None          # (DMT_INDEX_GROUP=0: everything on the default communicator -- the id exchange then queues behind gradient collectives)
    # ("force": a second group on ANY backend -- the soak test of tests/test_gpu_dp.py runs the two-communicator schedule over gloo)
    world_pg = dist.distributed_c10d._get_default_group()
    if _index_group["world"] is not world_pg:          # (a new default group after destroy_process_group + init_process_group)
        _index_group["world"] = world_pg
        _index_group["group"] = dist.new_group(backend=dist.get_backend())
    return _index_group["group"]


def allreduce_dense_(flat_grads: torch.Tensor, async_op: bool = False, force: bool = False):
    """Sum the flat gradient arena over ranks in place (the mean is applied by the optimizer's grad_scale).
    `force`: issue the collective even in a one-rank group (exercises the real RCCL call on a one-GPU box)."""
    if world()[1] == 1 and not (force and dist.is_available() and dist.is_initialized()):
        return None
    return dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, async_op=async_op)


def _all_gather_cat(dst: torch.Tensor, loc: torch.Tensor, W: int, cap: int, group=None):
    if dist.get_backend() == "nccl" and hasattr(dist, "all_gather_into_tensor"):
        dist.all_gather_into_tensor(dst, loc, group=group)      # one RCCL all-gather straight into the rank-major buffer
        return
    parts = [torch.empty_like(loc) for _ in range(W)]
    dist.all_gather(parts, loc, group=group)
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


def _a2a(dst: torch.Tensor, src: torch.Tensor, recv_splits, send_splits, group=None):
    """all_to_all_single along dim 0 with uneven splits.  RCCL moves device tensors directly (one send/recv per peer: on the
    xGMI full mesh every pair has its own link); gloo (tests) only implements the CPU form, so device tensors are staged."""
    if dist.get_backend() != "nccl" and src.is_cuda:
        d_cpu = torch.empty(dst.shape, dtype=dst.dtype)
        dist.all_to_all_single(d_cpu, src.cpu(), recv_splits, send_splits, group=group)
        dst.copy_(d_cpu)
        return
    dist.all_to_all_single(dst, src, recv_splits, send_splits, group=group)


def owner_of(keys: torch.Tensor, W: int) -> torch.Tensor:
    """Owner rank of a global row id in the owner-reduce exchange: interleaved (row % W), so the hot low ids of a Zipf law
    spread evenly over the ranks."""
    return torch.remainder(keys, W)


def exchange_to_owners(keys: torch.Tensor, rows: torch.Tensor, n: int, transport_dtype=None, group_fn=None):
    """First half of the owner-reduce exchange.  keys [>= n] int32 distinct row ids of this rank, rows [>= n, D] their
    gradient rows.  Every (key, row) pair travels to rank key % W -- ONE all_to_all_single for the keys and one for the rows
    (bf16 on the wire when transport_dtype says so) -- and arrives concatenated in RANK ORDER, which keeps the order of
    summation of the reference's tower loop (run_dnn.py:45-80).  Returns (recv_keys [R], recv_rows [R, D]).
    group_fn(keys, owner, rows, transport_dtype) -> (keys grouped by owner, rows grouped and in wire format): optional device
    implementation of the stable grouping (the default is torch.sort + index_select, used on CPU)."""
    rank, W = world()
    dev = keys.device
    k = keys[:n]
    owner = owner_of(k, W)
    counts = torch.bincount(owner.long(), minlength=W)
    rows_mat = [torch.zeros_like(counts) for _ in range(W)]
    dist.all_gather(rows_mat, counts)
    M = torch.stack(rows_mat).cpu()                      # M[s, d] = pairs rank s sends to rank d  (the one host sync)
    send_splits, recv_splits = M[rank].tolist(), M[:, rank].tolist()
    if group_fn is not None:
        # device path (Trainer): one radix pass over the owner ids + one gather-and-round kernel (dmt_rows_permute)
        send_k, send_r = group_fn(k, owner, rows[:n], transport_dtype)
    else:
        _so, perm = torch.sort(owner, stable=True)       # per-owner slices, each still ascending in key
        send_k = k.index_select(0, perm)
        send_r = rows[:n].index_select(0, perm)
        if transport_dtype is not None and transport_dtype != send_r.dtype:
            send_r = send_r.to(transport_dtype)
    R = int(sum(recv_splits))
    recv_k = torch.empty((R,), dtype=keys.dtype, device=dev)
    recv_r = torch.empty((R, rows.shape[1]), dtype=send_r.dtype, device=dev)
    _a2a(recv_k, send_k.contiguous(), recv_splits, send_splits)
    _a2a(recv_r, send_r.contiguous(), recv_splits, send_splits)
    return recv_k, recv_r


def request_rows(keys: torch.Tensor, n: int):
    """Row-sharded tables, forward exchange, first half (the "all-to-all index exchange" of BASELINE configs[3]): this rank's n distinct
    global row ids travel to their owners (id % W).  Returns (perm, recv_keys, send_splits, recv_splits): perm[i] = index (into keys)
    of the i-th id in owner-grouped order -- the order in which the rows will come back --, recv_keys = the ids the other ranks want
    from this one, concatenated in rank order."""
    rank, W = world()
    dev = keys.device
    k = keys[:n]
    if W == 1:
        perm = torch.arange(n, device=dev)
        return perm, k, [n], [n]
    owner = owner_of(k, W)
    counts = torch.bincount(owner.long(), minlength=W)
    rows_mat = [torch.zeros_like(counts) for _ in range(W)]
    dist.all_gather(rows_mat, counts)
    M = torch.stack(rows_mat).cpu()                      # M[s, d] = ids rank s asks of rank d  (one host sync)
    send_splits, recv_splits = M[rank].tolist(), M[:, rank].tolist()
    _so, perm = torch.sort(owner, stable=True)
    send_k = k.index_select(0, perm).contiguous()
    recv_k = torch.empty((int(sum(recv_splits)),), dtype=keys.dtype, device=dev)
    _a2a(recv_k, send_k, recv_splits, send_splits)
    return perm, recv_k, send_splits, recv_splits


def return_rows(rows: torch.Tensor, n_back: int, send_splits, recv_splits):
    """Second half: the owner's rows [R, D] (in the order of request_rows' recv_keys) go back to the ranks that asked; returns
    [n_back, D] in this rank's owner-grouped request order."""
    rank, W = world()
    if W == 1:
        return rows
    out = torch.empty((n_back, rows.shape[1]), dtype=rows.dtype, device=rows.device)
    _a2a(out, rows.contiguous(), send_splits, recv_splits)   # what was received is now sent, and vice versa
    return out


def allgather_shards(keys: torch.Tensor, rows: torch.Tensor, m: int, invalid_key: int, transport_dtype=None):
    """Second half: every owner's reduced shard (m distinct keys, rows [>= m, D]) goes to every rank.  Shards are padded to the
    largest one (interleaved ownership keeps them within a few per cent of each other); padding slots carry `invalid_key`,
    which the optimizer kernels skip.  Returns (all_keys [W*cap], all_rows [W*cap, D], cap)."""
    rank, W = world()
    dev = keys.device
    # m may be a device tensor (the shard's distinct-row count as the segment kernel left it): the shard sizes of all ranks and
    # this rank's own m then cost ONE host sync together
    cnt = m.reshape(1).to(torch.int64) if isinstance(m, torch.Tensor) else torch.tensor([m], dtype=torch.int64, device=dev)
    cnts = [torch.zeros_like(cnt) for _ in range(W)]
    dist.all_gather(cnts, cnt)
    c_host = torch.stack(cnts).reshape(-1).cpu()
    m = int(c_host[rank])
    cap = max(1, int(c_host.max()))
    k_loc = torch.full((cap,), invalid_key, dtype=keys.dtype, device=dev)
    k_loc[:m] = keys[:m]
    if rows.shape[0] >= cap:
        r_loc = rows[:cap]
    else:
        r_loc = torch.zeros((cap, rows.shape[1]), dtype=rows.dtype, device=dev)
        r_loc[:m] = rows[:m]
    if transport_dtype is not None and transport_dtype != r_loc.dtype:
        r_loc = r_loc.to(transport_dtype)
    all_k = torch.empty((W * cap,), dtype=keys.dtype, device=dev)
    all_r = torch.empty((W * cap, rows.shape[1]), dtype=r_loc.dtype, device=dev)
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


def _parse_cpulist(text: str):
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.extend(range(int(a), int(b or a) + 1))
    return out


def device_bdf(props) -> str:
    """sysfs name ('dddd:bb:dd.f') of a device from torch's properties: `pci_bus_id` is the BUS NUMBER (an int) in torch 2.10+rocm, a BDF
    string in some other builds."""
    bus = getattr(props, "pci_bus_id", None)
    if isinstance(bus, str):
        return bus.lower()
    if bus is None:
        return ""
    return "%04x:%02x:%02x.0" % (int(getattr(props, "pci_domain_id", 0)), int(bus), int(getattr(props, "pci_device_id", 0)))


def pin_rank_to_cores(local_rank: int, local_world: int, device_index: int = None):
    """CPU plan of one rank of a one-process-per-GPU job: this process -- its Python launcher thread, the autograd thread and the native
    input stage's parser pool (threads inherit the mask) -- is confined to an equal share of the host's cores, taken from the NUMA node
    the rank's GPU hangs off when sysfs tells (8 ranks x (launcher + 32 parser threads) otherwise wander over all 256 cores of the host,
    and a parser thread that lands two sockets away from its page cache and its pinned output buffer pays for it on every record).
    Returns {"cores": [...], "numa_node": n or None}; a no-op ({}) where sched_setaffinity does not exist.  DMT_PIN_CORES=0 disables."""
    import os
    if os.environ.get("DMT_PIN_CORES", "1") == "0" or not hasattr(os, "sched_setaffinity") or local_world < 1:
        return {}
    allowed = sorted(os.sched_getaffinity(0))
    node, node_cores = None, None
    if device_index is not None and torch.cuda.is_available():
        try:
            bdf = device_bdf(torch.cuda.get_device_properties(device_index))
            path = "/sys/bus/pci/devices/%s/numa_node" % bdf
            if bdf and os.path.exists(path):
                n = int(open(path).read().strip())
                if n >= 0:
                    cl = _parse_cpulist(open("/sys/devices/system/node/node%d/cpulist" % n).read())
                    node, node_cores = n, [c for c in cl if c in set(allowed)]
        except Exception as e:          # (never fatal: the plan below falls back to a plain slice -- but said aloud, not swallowed)
            import warnings
            warnings.warn("pin_rank_to_cores: NUMA node of device %s not found (%s: %s); the rank keeps a plain slice of the host's cores"
                          % (device_index, type(e).__name__, e))
            node, node_cores = None, None
    share = max(1, len(allowed) // local_world)
    if node_cores and len(node_cores) >= share:
        # the ranks whose GPUs share this node split ITS cores: position of this rank among them = its index modulo ranks per node
        per_node = max(1, len(node_cores) // share)
        k = local_rank % per_node
        cores = node_cores[k * share:(k + 1) * share]
    else:
        cores = allowed[local_rank * share:(local_rank + 1) * share]
    if not cores:
        return {}
    os.sched_setaffinity(0, cores)
    return {"cores": cores, "numa_node": node}


"""Data-parallel exchange on CPU (gloo, world_size 2): the dense all-reduce and the rank-ordered sparse all-gather
reproduce the reference's average_gradients (run_dnn.py:45-80) -- mean over towers of per-tower gradients."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cikm2020_dmt_amd import parallel
from cikm2020_dmt_amd.data_feed.synthetic import make_batch
from cikm2020_dmt_amd.variables import VariableStore
from oracle import dmt_oracle as O
from oracle import dmt_oracle_torch as OT
from tests.util import small_specs


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _rank_grads(rank, so, sp, P):
    inputs, mask, _ = make_batch(sp, 5, seed=40 + rank, lengths="ragged", weights="random")
    loss, _lg, G = OT.loss_and_grads(P, inputs, mask, so)
    return loss, G


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    so, sp = small_specs()
    P = O.init_params(so, seed=1)
    store = VariableStore(sp, "cpu", torch.float32, seed=0)
    loss, G = _rank_grads(rank, so, sp, P)
    # ---- dense: flat arena all-reduce, mean applied as grad_scale = 1/world
    with torch.no_grad():
        for tf_name, v in store.views.items():
            store.leaf[v.leaf].grad[v.index].copy_(torch.as_tensor(G[tf_name], dtype=torch.float32))
    work = parallel.allreduce_dense_(store.grads, async_op=True)
    work.wait()
    dense = {k: v / world for k, v in store.grad_dict().items()}
    # ---- sparse: this rank's distinct touched rows (global ids) and their gradient rows
    D = max(t.shape[1] for t in store.table.values())
    keys, rows = [], []
    for name, (base, nrows) in sorted(store.table_rows.items(), key=lambda kv: kv[1][0]):
        g = G[name]
        touched = np.nonzero(np.abs(g).sum(1) > 0)[0]
        for r in touched:
            keys.append(base + r)
            row = np.zeros(D, np.float32); row[: g.shape[1]] = g[r]
            rows.append(row)
    n = len(keys)
    k_t = torch.tensor(np.array(keys + [0] * 3, dtype=np.int32))          # buffers may be larger than n
    r_t = torch.tensor(np.array(rows + [np.zeros(D, np.float32)] * 3))
    all_k, all_r, cap = parallel.allgather_sparse(k_t, r_t, n, store.total_rows)
    assert all_k.numel() == world * cap
    assert int((all_k[rank * cap: rank * cap + n] != k_t[:n]).sum()) == 0            # rank-major placement
    assert bool((all_k[rank * cap + n: (rank + 1) * cap] == store.total_rows).all())  # padding slots are invalid keys
    order = np.argsort(all_k.numpy(), kind="stable")                                  # what dmt_sort_pairs does on the GPU
    ks, rs = all_k.numpy()[order], all_r.numpy()[order]
    merged = {}
    for k, r in zip(ks, rs):
        if k >= store.total_rows:
            break
        merged[int(k)] = merged.get(int(k), 0) + r.astype(np.float64)
    mloss = parallel.mean_scalar(torch.tensor(loss))
    q.put((rank, float(mloss), {k: v for k, v in dense.items() if k.endswith("weights") or k.endswith("kernel")},
           {k: v / world for k, v in merged.items()}, n))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_exchange_matches_mean_of_tower_gradients():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p_ in procs:
        p_.start()
    res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
    for p_ in procs:
        p_.join(60)
        assert p_.exitcode == 0
    so, sp = small_specs()
    P = O.init_params(so, seed=1)
    store = VariableStore(sp, "cpu", torch.float32, seed=0)
    L, Gs = zip(*[_rank_grads(r, so, sp, P) for r in range(world)])
    assert abs(res[0][1] - np.mean(L)) < 1e-6 and res[0][1] == res[1][1]   # float32 scalar on the wire
    for r in range(world):
        dense, merged = res[r][2], res[r][3]
        for name, g in dense.items():
            ref = (Gs[0][name] + Gs[1][name]) / world
            assert np.abs(g - ref).max() < 1e-6 * max(1.0, np.abs(ref).max()), name
        for name, (base, nrows) in store.table_rows.items():
            ref = (Gs[0][name] + Gs[1][name]) / world
            for row in np.nonzero(np.abs(ref).sum(1) > 0)[0]:
                got = merged[base + int(row)][: ref.shape[1]]
                assert np.abs(got - ref[row]).max() < 1e-6 * max(1.0, np.abs(ref[row]).max())
    assert res[0][3].keys() == res[1][3].keys()        # every rank ends with the identical merged row set


def test_single_process_paths_are_noops():
    assert parallel.world() == (0, 1)
    g = torch.ones(4)
    assert parallel.allreduce_dense_(g) is None
    k, r, cap = parallel.allgather_sparse(torch.tensor([3, 9, 0], dtype=torch.int32), torch.ones((3, 2)), 2, 100)
    assert cap == 2 and k.tolist() == [3, 9] and r.shape == (2, 2)
    assert float(parallel.mean_scalar(torch.tensor(2.5))) == 2.5


# ------------------------------------------------------------------------------------------------ owner-reduce exchange
def _rank_rows(rank, total, D):
    rng = np.random.default_rng(900 + rank)
    n = int(rng.integers(40, 80))
    hot = np.arange(12)                                                       # rows every rank touches (the Zipf head)
    keys = np.unique(np.concatenate([hot, rng.choice(total, size=n, replace=False)])).astype(np.int32)
    rows = rng.standard_normal((len(keys), D)).astype(np.float32)
    return keys, rows


def _owner_worker(rank, world, port, q, total, D):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    keys, rows = _rank_rows(rank, total, D)
    n = len(keys)
    k_t = torch.tensor(np.concatenate([keys, np.zeros(5, np.int32)]))         # buffers may be larger than n
    r_t = torch.tensor(np.concatenate([rows, np.zeros((5, D), np.float32)]))
    rk, rr = parallel.exchange_to_owners(k_t, r_t, n)
    assert bool((torch.remainder(rk, world) == rank).all())                   # only rows this rank owns arrive
    # second-level reduce in arrival (= rank) order: what merge_gathered's stable sort + segment reduce does on the GPU
    order = np.argsort(rk.numpy(), kind="stable")
    ks, rs = rk.numpy()[order], rr.numpy()[order]
    uk, start = np.unique(ks, return_index=True)
    red = np.add.reduceat(rs.astype(np.float64), start, axis=0).astype(np.float32) if len(ks) else np.zeros((0, D), np.float32)
    all_k, all_r, cap = parallel.allgather_shards(torch.tensor(uk.astype(np.int32)), torch.tensor(red), len(uk), total)
    assert all_k.numel() == world * cap and bool((all_k[rank * cap + len(uk): (rank + 1) * cap] == total).all())
    valid = all_k < total
    q.put((rank, all_k[valid].numpy(), all_r[valid].numpy(), int(rk.numel())))
    dist.barrier()
    dist.destroy_process_group()


def test_three_rank_owner_reduce_exchange_sums_every_row_once():
    world, total, D = 3, 1000, 8
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_owner_worker, args=(r, world, port, q, total, D)) for r in range(world)]
    for p_ in procs:
        p_.start()
    res = sorted([q.get(timeout=300) for _ in range(world)], key=lambda t: t[0])
    for p_ in procs:
        p_.join(60)
        assert p_.exitcode == 0
    want = {}
    for r in range(world):
        keys, rows = _rank_rows(r, total, D)
        for k, row in zip(keys, rows):
            want[int(k)] = want.get(int(k), 0) + row.astype(np.float64)
    for rank, ks, rs, received in res:
        assert sorted(ks.tolist()) == sorted(want)                             # the union, every row exactly once
        for k, row in zip(ks, rs):
            assert np.abs(row - want[int(k)]).max() < 1e-5
        assert np.array_equal(ks, res[0][1]) and np.array_equal(rs, res[0][2])  # identical on every rank (same order too)
    assert sum(t[3] for t in res) == sum(len(_rank_rows(r, total, D)[0]) for r in range(world))   # each pair travelled once


def _rows_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(10 + rank)
    total = 1000
    table = (np.arange(total, dtype=np.float32)[:, None] * 4 + np.arange(4, dtype=np.float32)[None, :])   # row g = [4g .. 4g+3]
    mine = table[rank::world]                                      # the shard this rank owns: local row l = global row l*W + rank
    n = int(rng.integers(0, 60)) if rank else 57                   # ragged request sizes, possibly empty
    keys = np.sort(rng.choice(total, size=n, replace=False)).astype(np.int32)
    k_t = torch.tensor(np.concatenate([keys, np.zeros(5, np.int32)]))
    perm, recv_k, ss, rs = parallel.request_rows(k_t, n)
    rk = recv_k.numpy()
    assert (rk % world == rank).all()                              # only ids this rank owns arrive
    rows = torch.tensor(mine[rk // world].reshape(len(rk), 4))
    back = parallel.return_rows(rows, n, ss, rs)
    want = table[keys[perm.numpy()]] if n else np.zeros((0, 4), np.float32)
    ok = back.shape == (n, 4) and np.array_equal(back.numpy(), want)
    # owner-grouped order: perm groups by owner, ascending ids inside a group
    own = keys[perm.numpy()] % world
    ok = ok and bool((np.diff(own) >= 0).all())
    q.put((rank, bool(ok), n))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_row_request_and_return_exchange(world):
    """Row-sharded tables (BASELINE configs[3]): ids to the owners (id % W), rows back, ragged and empty requests."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rows_worker, args=(r, world, port, q)) for r in range(world)]
    for p_ in procs:
        p_.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p_ in procs:
        p_.join(60)
        assert p_.exitcode == 0
    assert all(ok for (_r, ok, _n) in res), res

"""Row-sharded embedding tables (BASELINE configs[3]): rank r holds the rows with global id % W == r; the batch's
distinct ids travel to the owners, the rows come back into a per-batch row cache, gradient rows travel to the owners, only
owners run Adam.  The result must be the replicated layout's, bit for bit in fp32 (same pairs, same rank order of summation)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import dmt_oracle as O
from cikm2020_dmt_amd.data_feed.synthetic import make_batch
from tests.util import small_specs

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q, steps, layout, bf16, backend="gloo"):
    import torch.distributed as dist
    from cikm2020_dmt_amd import ops
    from cikm2020_dmt_amd.train import Trainer
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if backend == "nccl":
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    so, sp = small_specs()
    P = O.init_params(so, seed=5)
    if os.environ.get("DMT_TEST_POLLUTE", "nan") != "off":
        # the caching allocator's free blocks are filled with NaNs first: a kernel that reads memory nobody wrote, or a buffer that
        # lands elsewhere than in a fresh process, shows up here (this is how the duplicate-row replay race of fetch_rows was caught:
        # inside the whole suite the test failed in two runs out of five, alone never)
        fill = float(os.environ.get("DMT_TEST_POLLUTE", "nan"))
        junk = [torch.full((1 << 26,), fill, device="cuda:0") for _ in range(8)]
        junk += [torch.full((n,), fill, device="cuda:0") for n in (7, 64, 300, 4096, 20000, 70000, 1 << 20) for _ in range(6)]
        torch.cuda.synchronize()
        del junk
    if not bf16:
        # bit-for-bit claims hold in deterministic mode: in the default mode partial sums that meet in one element are combined with
        # fp32 atomics, and their order follows the timing of the kernels (two processes and three stream lanes share the GPU here)
        ops.set_deterministic(True)
    tr = Trainer(sp, device="cuda:0", compute_dtype=torch.bfloat16 if bf16 else torch.float32, init=False, dropout=False,
                 table_layout=layout, force_dp=(backend == "nccl"))
    tr.store.load_state(P)
    if layout == "sharded" and world > 1:
        for name, t in tr.store.table.items():      # this process really holds ~1/W of every table
            rows = tr.store.tables[name].shape[0]
            assert t.shape[0] == (rows + world - 1) // world
    losses = []
    for s in range(steps):
        # steps 0,1 touch different rows than step 2.. (different seeds): the owners' lazy Adam has pending updates to replay
        inputs, mask, _ = make_batch(sp, 7, seed=900 + 10 * s + rank, lengths="ragged", weights="random")
        losses.append(float(tr.train_step(tr.make_batch(inputs, mask))))
    inputs, mask, _ = make_batch(sp, 5, seed=77, lengths="ragged", weights="random")
    pc, po = tr.predict(tr.make_batch(inputs, mask))
    tr.opt.flush_tables()
    torch.cuda.synchronize()
    sd = tr.store.state_dict()                      # (sharded: a collective -- every rank gathers the whole tables)
    q.put((rank, losses, sd, pc.float().cpu().numpy(), po.float().cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def _run(world, steps, layout, bf16=False, backend="gloo"):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, steps, layout, bf16, backend)) for r in range(world)]
    for p_ in procs:
        p_.start()
    res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
    for p_ in procs:
        p_.join(120)
        assert p_.exitcode == 0
    return res


def test_sharded_tables_equal_replicated_tables_bitwise(cuda):
    """fp32: 2 ranks with row-sharded tables == 2 ranks with replicated tables, after 4 steps: every variable (gathered tables
    included), every loss and the predict scores, bit for bit (same pairs at the same owners, same order of summation)."""
    world = 2
    a = _run(world, 4, "sharded")
    b = _run(world, 4, "replicated")
    for r in range(world):
        assert a[r][1] == b[r][1]
        for k in b[0][2]:
            assert a[r][2][k].shape == b[0][2][k].shape, k
            assert np.array_equal(a[r][2][k].view(np.uint32), b[0][2][k].view(np.uint32)), (k, r)
        assert np.array_equal(a[r][3], b[r][3]) and np.array_equal(a[r][4], b[r][4])


def test_sharded_tables_three_ranks_padded_shards(cuda):
    """world 3 does not divide the table sizes: every table is padded to a multiple of 3 global row ids, which moves the row bases and
    with them the owner (id % 3) of a row relative to the replicated layout's exchange.  The same pairs are then summed at a different
    owner, at different positions of its segmented reduce -- (a+b)+c against a+(b+c) -- so equality is to rounding, not bitwise: after
    ONE step every variable agrees to 1e-7; later steps stay within what Adam makes of such a difference (a sign flip of a ~1e-9
    gradient moves a parameter by lr = 1e-3)."""
    a1, b1 = _run(3, 1, "sharded"), _run(3, 1, "replicated")
    for r in range(3):
        for k in b1[0][2]:
            assert a1[r][2][k].shape == b1[0][2][k].shape, k
            assert np.abs(a1[r][2][k] - b1[0][2][k]).max() <= 1e-7, (k, r)
        assert np.array_equal(a1[r][2]["embedding_trans/Sku/embedding"], a1[0][2]["embedding_trans/Sku/embedding"])
    a, b = _run(3, 4, "sharded"), _run(3, 4, "replicated")
    assert np.abs(np.array(a[0][1]) - np.array(b[0][1])).max() < 5e-3
    for k in b[0][2]:
        assert np.abs(a[0][2][k] - b[0][2][k]).max() < 4e-3, k
        for r in (1, 2):
            assert np.array_equal(a[r][2][k], a[0][2][k]), k          # the gathered tables are the same on every rank
    assert np.abs(a[0][3] - b[0][3]).max() < 2e-3


def test_sharded_tables_bf16_mode_ranks_agree_and_track_replicated(cuda):
    """bf16 mode: gradient rows travel to the owners as bf16 in both layouts; the replicated layout then rounds the REDUCED rows to bf16
    once more for the shard all-gather, the sharded layout applies the owner's fp32 sums directly -- so the two agree to bf16
    rounding of the gradient (a few Adam steps of 1e-3), not bitwise."""
    a = _run(2, 3, "sharded", bf16=True)
    b = _run(2, 3, "replicated", bf16=True)
    assert np.abs(np.array(a[0][1]) - np.array(b[0][1])).max() < 3e-2
    for k in a[0][2]:
        assert np.array_equal(a[0][2][k], a[1][2][k]), k
        assert np.abs(a[0][2][k] - b[0][2][k]).max() < 5e-3, k


def test_sharded_layout_one_rank_rccl_group_equals_plain_step(cuda):
    """The sharded step's collectives (all_gather of counts, all_to_all_single of ids / rows / gradient rows) through REAL RCCL in a
    one-rank group, against the plain replicated one-GPU step: bit-identical in fp32."""
    a = _run(1, 3, "sharded", backend="nccl")
    b = _run(1, 3, "replicated", backend="gloo")
    assert a[0][1] == b[0][1]
    for k in b[0][2]:
        assert np.array_equal(a[0][2][k].view(np.uint32), b[0][2][k].view(np.uint32)), k


def test_sharded_forward_requires_the_row_fetch(cuda):
    from cikm2020_dmt_amd.train import Trainer
    so, sp = small_specs()
    tr = Trainer(sp, device="cuda:0", init=True, dropout=False, table_layout="sharded")
    inputs, mask, _ = make_batch(sp, 4, seed=1)
    b = tr.make_batch(inputs, mask)
    with pytest.raises(RuntimeError, match="sync_rows"):
        tr.engine.inference(b)


def _ckpt_worker(rank, world, port, q, model_path):
    import torch.distributed as dist
    from cikm2020_dmt_amd import checkpoint as CK
    from cikm2020_dmt_amd.train import Trainer
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    so, sp = small_specs()
    tr = Trainer(sp, device="cuda:0", compute_dtype=torch.float32, seed=4, dropout=False, table_layout="sharded")
    for s in range(2):
        inputs, mask, _ = make_batch(sp, 7, seed=500 + 10 * s + rank, lengths="ragged", weights="random")
        tr.train_step(tr.make_batch(inputs, mask))
    calls = []
    orig = tr.store.full_table
    tr.store.full_table = lambda name: (calls.append(name), orig(name))[1]
    path = CK.save(tr, model_path)                       # every rank calls it; no table is gathered
    assert calls == [], calls
    tr.store.full_table = orig
    files = sorted(os.listdir(model_path))
    sd = tr.store.state_dict()                           # (collective gather: the test's view of the whole tables)
    # (a) same sharding: every rank reads its own shard file
    t2 = Trainer(sp, device="cuda:0", compute_dtype=torch.float32, seed=99, dropout=False, table_layout="sharded")
    assert CK.restore(t2, model_path) == 2
    sd2 = t2.store.state_dict()
    # (b) into replicated tables (re-assembled on the host, one table at a time)
    t3 = Trainer(sp, device="cuda:0", compute_dtype=torch.float32, seed=77, dropout=False, table_layout="replicated")
    assert CK.restore(t3, model_path) == 2
    sd3 = t3.store.state_dict()
    ok = all(np.array_equal(sd[k], sd2[k]) and np.array_equal(sd[k], sd3[k]) for k in sd)
    q.put((rank, files, ok, os.path.basename(path)))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_checkpoint_writes_shards_without_gathering_and_restores_into_any_layout(cuda, tmp_path):
    """checkpoint.save with row-sharded tables (round-2 advice): no rank gathers a whole table (full_table is never called), every rank
    writes only its own shard file, rank 0 the dense variables and -- after a barrier -- the DONE marker; restore reads the shards back
    into the same sharding or re-assembles them for another layout."""
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    model_path = str(tmp_path / "model") + os.sep
    procs = [ctx.Process(target=_ckpt_worker, args=(r, world, port, q, model_path)) for r in range(world)]
    for p_ in procs:
        p_.start()
    res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
    for p_ in procs:
        p_.join(120)
        assert p_.exitcode == 0
    for (_r, files, ok, name) in res:
        assert ok and name == "model.ckpt-2.npz"
        assert files == ["model.ckpt-2.npz", "model.ckpt-2.shard-00000-of-00002.npz", "model.ckpt-2.shard-00001-of-00002.npz", "step-2.model.DONE"], files

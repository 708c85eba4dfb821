"""Data-parallel train step with TWO ranks (both on the one visible GPU, gloo transport over device tensors): the
exchange + merge + optimizer path of Trainer.train_step against the oracle's mean-of-tower gradients + TF-Adam.
(RCCL itself needs >= 2 GPUs; the rank logic, sparse merge kernels and optimizer are identical.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import dmt_oracle as O
from oracle import dmt_oracle_torch as OT
from cikm2020_dmt_amd.data_feed.synthetic import make_batch
from tests.util import small_specs

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q, steps, bf16=False, dp_exchange="owner", wgrad_min_rows=None):
    import torch.distributed as dist
    from cikm2020_dmt_amd import ops as _ops
    from cikm2020_dmt_amd.train import Trainer
    if wgrad_min_rows is not None:
        _ops.WGRAD320_MIN_ROWS = wgrad_min_rows          # every weight gradient with M >= this many rows counts as "long-row"
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    so, sp = small_specs()
    P = O.init_params(so, seed=5)
    if os.environ.get("DMT_TEST_POLLUTE", "nan") != "off":        # (NaN-filled free blocks: tests/test_gpu_sharded.py)
        fill = float(os.environ.get("DMT_TEST_POLLUTE", "nan"))
        junk = [torch.full((1 << 26,), fill, device="cuda:0") for _ in range(4)]
        junk += [torch.full((n,), fill, device="cuda:0") for n in (7, 64, 300, 4096, 20000, 70000, 1 << 20) for _ in range(6)]
        torch.cuda.synchronize()
        del junk
    if not bf16:
        from cikm2020_dmt_amd import ops
        ops.set_deterministic(True)     # (bit-for-bit comparisons BETWEEN runs: the default mode's fp32 atomics follow kernel timing)
    tr = Trainer(sp, device="cuda:0", compute_dtype=torch.bfloat16 if bf16 else torch.float32, init=False, dp_exchange=dp_exchange, dropout=False)
    tr.store.load_state(P)
    losses = []
    tr.diag = {}                            # HIP-event spans of the step's phases (bench.py: step_phases_ms)
    for s in range(steps):
        inputs, mask, _ = make_batch(sp, 6, seed=700 + 10 * s + rank, lengths="ragged", weights="random")
        losses.append(float(tr.train_step(tr.make_batch(inputs, mask))))
    assert tr.early_allreduce_used          # the MMoE / tower slice was reduced from the dL/dz hook, during backward
    torch.cuda.synchronize()
    want = {"index_plane_sort", "allreduce_head", "allreduce_tail", "optimizer", "exchange_ids", "exchange_rows_a2a",
            "exchange_rows_allgather"} if dp_exchange == "owner" else {"index_plane_sort"}      # (the one-shot exchange form is not instrumented)
    missing = want - set(tr.diag)
    assert not missing, ("phases the data-parallel step must report", missing, sorted(tr.diag))
    for key, ent in tr.diag.items():
        assert all(e0.elapsed_time(e1) >= 0.0 for (e0, e1) in ent), key
    tr.diag = None
    tr.opt.flush_tables()
    torch.cuda.synchronize()
    q.put((rank, losses, tr.store.state_dict()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("wgrad_min_rows", [None, 4])
def test_two_rank_train_steps_match_oracle(cuda, wgrad_min_rows):
    """wgrad_min_rows = 4: with a per-rank batch of 6 EVERY weight gradient of the step counts as long-row (as the MMoE layer 0 and the
    towers' hidden layers do at a per-rank batch >= 16384): those of the arena's tail must still be complete when the early
    all-reduce fires from the dL/dz hook (round-2 advice: they were collected, missed the all-reduce, and the replicas diverged)."""
    world, steps = 2, 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, steps, False, "owner", wgrad_min_rows)) for r in range(world)]
    for p_ in procs:
        p_.start()
    res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
    for p_ in procs:
        p_.join(120)
        assert p_.exitcode == 0
    # every rank holds the identical replica
    for k in res[0][2]:
        assert np.array_equal(res[0][2][k], res[1][2][k]), k
    # oracle: gradient = mean over towers of the per-tower gradients (run_dnn.py:45-80), one Adam step per global step
    so, sp = small_specs()
    P = {k: v.copy() for k, v in O.init_params(so, seed=5).items()}
    adam = O.TFAdam(lr=1e-3)
    ref_losses = []
    for s in range(steps):
        Gs, Ls = [], []
        for r in range(world):
            inputs, mask, _ = make_batch(sp, 6, seed=700 + 10 * s + r, lengths="ragged", weights="random")
            l, _lg, G = OT.loss_and_grads(P, inputs, mask, so)
            Gs.append(G); Ls.append(l)
        adam.apply(P, {k: (Gs[0][k] + Gs[1][k]) / world for k in Gs[0]})
        ref_losses.append(np.mean(Ls))
    assert np.abs(np.array(res[0][1]) - np.array(ref_losses)).max() < 1e-4
    got = res[0][2]
    total = sum(P[k].size for k in P)
    n_off = sum(int((np.abs(got[k] - P[k]) > 2e-5).sum()) for k in P)
    worst = max(float(np.abs(got[k] - P[k]).max()) for k in P)
    assert n_off <= 1e-4 * total and worst < 5e-4, (n_off, total, worst)


def test_two_rank_bf16_mode_replicas_identical_and_close_to_oracle(cuda):
    """bf16 mode: the per-rank gradient rows travel as bf16 (dmt_rows_reduce_bf16); both ranks must still hold the identical
    replica, and the result stays inside the bf16 tolerance of the fp64 oracle."""
    world, steps = 2, 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, steps, True)) for r in range(world)]
    for p_ in procs:
        p_.start()
    res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
    for p_ in procs:
        p_.join(120)
        assert p_.exitcode == 0
    for k in res[0][2]:
        assert np.array_equal(res[0][2][k], res[1][2][k]), k
    so, sp = small_specs()
    P = {k: v.copy() for k, v in O.init_params(so, seed=5).items()}
    adam = O.TFAdam(lr=1e-3)
    ref_losses = []
    for s in range(steps):
        Gs, Ls = [], []
        for r in range(world):
            inputs, mask, _ = make_batch(sp, 6, seed=700 + 10 * s + r, lengths="ragged", weights="random")
            l, _lg, G = OT.loss_and_grads(P, inputs, mask, so)
            Gs.append(G); Ls.append(l)
        adam.apply(P, {k: (Gs[0][k] + Gs[1][k]) / world for k in Gs[0]})
        ref_losses.append(np.mean(Ls))
    assert np.abs(np.array(res[0][1]) - np.array(ref_losses)).max() < 3e-2     # bf16 forward tolerance on the loss
    got = res[0][2]
    # one Adam step moves every touched parameter by ~lr regardless of the gradient's magnitude: after 2 steps |dp| <= ~2e-3
    worst = max(float(np.abs(got[k] - P[k]).max()) for k in P)
    assert worst < 5e-3, worst


def _run(world, steps, bf16, dp_exchange):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, steps, bf16, dp_exchange)) for r in range(world)]
    for p_ in procs:
        p_.start()
    res = sorted([q.get(timeout=600) for _ in range(world)], key=lambda t: t[0])
    for p_ in procs:
        p_.join(120)
        assert p_.exitcode == 0
    return res


def test_owner_reduce_exchange_equals_allgather_exchange(cuda):
    """The two transports of the embedding-gradient rows (owner-reduce: all_to_all + all-gather of reduced shards; and the
    one-step all-gather with a full second-level reduce on every rank) sum the same pairs in the same rank order: in fp32
    mode the replicas after two steps are bit-identical, on both ranks."""
    a = _run(2, 2, False, "owner")
    b = _run(2, 2, False, "allgather")
    for k in a[0][2]:
        assert np.array_equal(a[0][2][k], a[1][2][k]), k
        assert np.array_equal(a[0][2][k], b[0][2][k]), k
    assert a[0][1] == b[0][1]


def _nccl_world1(q):
    import torch.distributed as dist
    from cikm2020_dmt_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    g = torch.Generator(device="cpu").manual_seed(1)
    keys = torch.sort(torch.randperm(5000, generator=g)[:700]).values.to(torch.int32).to(dev)
    rows = torch.randn((700, 64), generator=g).to(dev)
    rk, rr = parallel.exchange_to_owners(keys, rows, 700, transport_dtype=torch.bfloat16)
    ok1 = torch.equal(rk, keys) and torch.equal(rr, rows.to(torch.bfloat16))
    ak, ar, cap = parallel.allgather_shards(rk, rr.float(), 700, 5000, transport_dtype=torch.bfloat16)
    ok2 = cap == 700 and torch.equal(ak, keys) and torch.equal(ar, rows.to(torch.bfloat16))
    flat = torch.ones(1000, device=dev)
    w = parallel.allreduce_dense_(flat, async_op=True)
    if w is not None:
        w.wait()
    torch.cuda.synchronize()
    q.put((bool(ok1), bool(ok2), float(flat.sum())))
    dist.destroy_process_group()


def test_rccl_accepts_the_exchange_calls_single_rank(cuda):
    """A one-GPU box cannot run two RCCL ranks; a ONE-rank RCCL communicator still runs the exact collectives of the N-rank
    step (all_to_all_single with uneven-split lists on int32 / bf16 device tensors, all_gather_into_tensor, all_reduce), so
    argument / dtype problems surface here rather than in the 8-GPU run."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p_ = ctx.Process(target=_nccl_world1, args=(q,))
    p_.start()
    ok1, ok2, s = q.get(timeout=300)
    p_.join(60)
    assert p_.exitcode == 0
    assert ok1 and ok2 and s == 1000.0


def _nccl_world1_step(q, bf16):
    import torch.distributed as dist
    from cikm2020_dmt_amd.train import Trainer
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    so, sp = small_specs()
    P = O.init_params(so, seed=5)
    if not bf16:
        from cikm2020_dmt_amd import ops
        ops.set_deterministic(True)     # (bit-for-bit comparisons BETWEEN runs: the default mode's fp32 atomics follow kernel timing)
    states = []
    for force, ahead in ((False, False), (True, False), (True, True)):
        tr = Trainer(sp, device="cuda:0", compute_dtype=torch.bfloat16 if bf16 else torch.float32, init=False, force_dp=force, dropout=False)
        tr.store.load_state(P)
        bs = []
        for s_ in range(3):
            inputs, mask, _ = make_batch(sp, 9, seed=300 + s_, lengths="ragged", weights="random")
            bs.append(tr.make_batch(inputs, mask))
        for s_ in range(3):
            # ahead: the id exchange of batch s + 1 (second communicator, index stream) is issued inside step s
            tr.train_step(bs[s_], prefetch=bs[s_ + 1] if (ahead and s_ < 2) else None)
            if ahead and s_ < 2:
                assert "xplan" in bs[s_ + 1]._prep
        if force:
            assert tr.early_allreduce_used
        tr.opt.flush_tables()
        torch.cuda.synchronize()
        states.append(tr.store.state_dict())
    worst = max(float(np.abs(states[0][k] - states[1][k]).max()) for k in states[0])
    same = all(np.array_equal(states[0][k], states[1][k]) for k in states[0])
    same = same and all(np.array_equal(states[1][k], states[2][k]) for k in states[0]) if not bf16 else same
    if bf16:
        worst = max(worst, max(float(np.abs(states[1][k] - states[2][k]).max()) for k in states[0]))
    q.put((same, worst))
    dist.destroy_process_group()


def _nccl_world1_overlap(q):
    import torch.distributed as dist
    from cikm2020_dmt_amd import ops
    from cikm2020_dmt_amd import spec as S
    from cikm2020_dmt_amd.train import Trainer
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    ops.set_deterministic(True)
    if os.environ.get("DMT_TEST_POLLUTE", "nan") != "off":        # (NaN-filled free blocks: tests/test_gpu_sharded.py)
        fill = float(os.environ.get("DMT_TEST_POLLUTE", "nan"))
        junk = [torch.full((1 << 26,), fill, device="cuda:0") for _ in range(8)]
        junk += [torch.full((n,), fill, device="cuda:0") for n in (7, 64, 300, 4096, 20000, 70000, 1 << 20) for _ in range(6)]
        torch.cuda.synchronize()
        del junk
    sp = S.scaled_spec(S.e64_spec(), {"Sku": 20000, "Brand": 3000, "Shopid": 3000, "Cid3": 1200})
    states, deferred = [], []
    for force, overlap, layout in ((False, False, "replicated"), (True, True, "replicated"), (True, False, "replicated"), (True, True, "sharded")):
        tr = Trainer(sp, device="cuda:0", compute_dtype=torch.float32, seed=3, force_dp=force, dropout=True, table_layout=layout)
        tr.overlap_wgrads = overlap
        tr.n_deferred = 0
        bs = []
        for s_ in range(3):
            inputs, mask, _ = make_batch(sp, 512, seed=400 + s_, lengths="ragged", weights="random")     # 512 x 50 rows >= 16384: deferred
            bs.append(tr.make_batch(inputs, mask))
        for s_ in range(3):
            tr.train_step(bs[s_], prefetch=bs[s_ + 1] if (force and s_ < 2) else None)
        deferred.append(tr.n_deferred)
        tr.opt.flush_tables()
        torch.cuda.synchronize()
        states.append(tr.store.state_dict())
    same = [all(np.array_equal(states[0][k], st[k]) for k in states[0]) for st in states[1:]]
    q.put((same, deferred))
    dist.destroy_process_group()


def test_weight_gradients_beside_the_row_exchange_change_nothing(cuda):
    """Data-parallel step with the long-row weight gradients collected during backward and launched while the gradient rows are on
    the links (first half beside the all_to_all, second half beside the all-gather): in deterministic fp32 mode three steps through a
    one-rank RCCL group end BIT-IDENTICAL to the plain one-GPU step -- with the overlap, without it, and with row-sharded tables."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p_ = ctx.Process(target=_nccl_world1_overlap, args=(q,))
    p_.start()
    same, deferred = q.get(timeout=600)
    p_.join(60)
    assert p_.exitcode == 0
    assert same == [True, True, True], same
    assert deferred[1] >= 6 and deferred[3] >= 6 and deferred[2] == 0, deferred


@pytest.mark.parametrize("bf16", [False, True])
def test_full_train_step_through_one_rank_rccl_group(cuda, bf16):
    """The N-rank train step (early + late all-reduce, all_to_all to the owners, shard all-gather, bf16-row Adam with padding
    keys) pushed through REAL RCCL calls in a one-rank group: in fp32 mode the result must be bit-identical to the plain
    one-GPU step; in bf16 mode the only difference is the bf16 wire format of the (single-contribution) gradient rows."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p_ = ctx.Process(target=_nccl_world1_step, args=(q, bf16))
    p_.start()
    same, worst = q.get(timeout=300)
    p_.join(60)
    assert p_.exitcode == 0
    if bf16:
        assert worst < 4e-3, worst          # three Adam steps move a parameter by at most ~3e-3
    else:
        assert same, worst


def _halves_worker(rank, world, port, q, B):
    import torch.distributed as dist
    from cikm2020_dmt_amd.train import Trainer
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    so, sp = small_specs()
    P = O.init_params(so, seed=5)
    tr = Trainer(sp, device="cuda:0", compute_dtype=torch.float32, init=False, dropout=False)
    tr.store.load_state(P)
    from cikm2020_dmt_amd.data_feed.synthetic import slice_batch
    for s in range(2):
        inputs, mask, _ = make_batch(sp, B, seed=4100 + s, lengths="ragged", weights="random")
        h = B // world
        sub_in, sub_mask = slice_batch(sp, inputs, mask, rank * h, (rank + 1) * h)
        loss = tr.train_step(tr.make_batch(sub_in, sub_mask))
    tr.opt.flush_tables()
    torch.cuda.synchronize()
    q.put((rank, float(loss), tr.store.state_dict()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_half_batches_equal_one_rank_whole_batch(cuda):
    """run_dnn.py:45-87: the mean over towers of per-tower mean losses / gradients.  With equal tower batches this IS the mean over the
    whole batch, so 2 ranks x B/2 must train like 1 rank x B.  fp32 mode; equality is to summation-order rounding (the per-rank
    sums are formed first), which Adam can amplify to ~lr on elements whose gradient is ~0: after two steps all but a few 1e-4 of
    the parameters agree to 2e-5 and none moves further than 2 Adam steps."""
    from cikm2020_dmt_amd.train import Trainer
    B = 12
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_halves_worker, args=(r, 2, port, q, B)) for r in range(2)]
    for p_ in procs:
        p_.start()
    res = sorted([q.get(timeout=600) for _ in range(2)], key=lambda t: t[0])
    for p_ in procs:
        p_.join(120)
        assert p_.exitcode == 0
    so, sp = small_specs()
    P = O.init_params(so, seed=5)
    tr = Trainer(sp, device="cuda:0", compute_dtype=torch.float32, init=False, dropout=False)
    tr.store.load_state(P)
    for s in range(2):
        inputs, mask, _ = make_batch(sp, B, seed=4100 + s, lengths="ragged", weights="random")
        loss = tr.train_step(tr.make_batch(inputs, mask))
    tr.opt.flush_tables()
    one = tr.store.state_dict()
    assert abs(float(loss) - res[0][1]) < 2e-5
    total = sum(v.size for v in one.values())
    n_off = sum(int((np.abs(res[0][2][k] - one[k]) > 2e-5).sum()) for k in one)
    worst = max(float(np.abs(res[0][2][k] - one[k]).max()) for k in one)
    assert n_off <= 3e-4 * total and worst < 2.5e-3, (n_off, total, worst)


def test_bench_two_ranks_torchrun_on_one_device(cuda):
    """The driver's N > 1 launch line (python -m torch.distributed.run ... bench.py --gpus 2), on this one-GPU box: both ranks on
    cuda:0 (DMT_BENCH_ONE_DEVICE=1) over gloo (RCCL needs two devices).  The JSON line must come out of rank 0 with the aggregate
    over both ranks."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DMT_BENCH_ONE_DEVICE="1", DMT_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "256",
           "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["scaling"] == "weak" and j["config"]["global_batch"] == 512
    assert abs(j["value"] - 512 * 3 / (j["ms_per_step"] * 3e-3)) / j["value"] < 1e-3
    assert np.isfinite(j["final_loss"])
    # the sharded-table variant of the same launch
    cmd2 = cmd + ["--shard-tables"]
    cmd2[cmd2.index("--master-port") + 1] = str(_free_port())
    out2 = subprocess.run(cmd2, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert out2.returncode == 0, out2.stderr[-3000:]
    j2 = json.loads([l for l in out2.stdout.splitlines() if l.startswith("{")][0])
    assert "row-sharded" in j2["config"]["parallelism"] and np.isfinite(j2["final_loss"])


@pytest.mark.gpu
def test_bench_gpus_2_launches_its_own_ranks(cuda):
    """`python bench.py --gpus 2` with no launcher in front (no WORLD_SIZE in the environment): bench.py starts the two ranks itself
    (run_dnn.py:148-207's towers = ranks here) and the line says n_gpus 2 -- never a one-rank number under a --gpus 2 request."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DMT_BENCH_ONE_DEVICE="1", DMT_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "256", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["global_batch"] == 512 and np.isfinite(j["final_loss"])


def _soak_worker(rank, world, port, q, steps):
    import torch.distributed as dist
    from cikm2020_dmt_amd import parallel
    from cikm2020_dmt_amd.train import Trainer
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["DMT_INDEX_GROUP"] = "force"            # the index plane of batch i + 1 on its own communicator, inside step i
    dist.init_process_group("gloo", rank=rank, world_size=world)
    so, sp = small_specs()
    P = O.init_params(so, seed=5)
    tr = Trainer(sp, device="cuda:0", compute_dtype=torch.bfloat16, init=False, dropout=True, dropout_seed=3)
    tr.store.load_state(P)
    assert parallel.index_group() is not None          # (a second communicator really is in use)
    rng = np.random.default_rng(100 + rank)
    batches = []
    for s in range(8):                                  # ragged sizes differ per rank and step: the exchanged counts never repeat in lock step
        inputs, mask, _ = make_batch(sp, int(rng.integers(3, 9)), seed=900 + 10 * s + rank, lengths="ragged", weights="random")
        batches.append(tr.make_batch(inputs, mask))
    losses = []
    for s in range(steps):
        cur, nxt = batches[s % 8], batches[(s + 1) % 8]
        nxt._prep = None                                # a fresh index plane every step, prefetched while step s runs
        losses.append(float(tr.train_step(cur, prefetch=nxt)))
    tr.opt.flush_tables()
    torch.cuda.synchronize()
    sd = tr.store.state_dict()
    import hashlib
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(np.ascontiguousarray(sd[k]).tobytes())
    q.put((rank, losses[-5:], h.hexdigest(), bool(np.all(np.isfinite(losses)))))
    dist.barrier()
    dist.destroy_process_group()


def test_soak_three_ranks_two_communicators_200_steps(cuda):
    """DESIGN.md section 6 risk: the id exchange of batch i + 1 runs on a SECOND communicator while step i's gradient collectives are in
    flight on the first.  200 prefetching steps on three ranks (gloo over the one GPU; ragged per-rank batch sizes so the two planes'
    message sizes drift against each other): no hang or reordering (the test has a hard timeout), every loss finite, and the three
    replicas bit-identical at the end -- a collective matched against the wrong partner would break exactly that."""
    world, steps = 3, 200
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_soak_worker, args=(r, world, port, q, steps)) for r in range(world)]
    for p_ in procs:
        p_.start()
    try:
        res = sorted([q.get(timeout=900) for _ in range(world)], key=lambda t: t[0])
    finally:
        for p_ in procs:
            p_.join(60)
            if p_.is_alive():
                p_.kill()
    assert all(p_.exitcode == 0 for p_ in procs)
    assert all(r[3] for r in res)
    assert res[0][2] == res[1][2] == res[2][2], "replicas diverged"

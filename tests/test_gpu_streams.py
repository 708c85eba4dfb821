"""One stream per behaviour sequence (DMTEngine.seq_streams): the three encoder / decoder pipelines of a step overlap on the GPU.
Nothing may change but the schedule: loss, outputs and every gradient against the single-stream run, repeatedly (a race between
streams would show up as an occasional large difference), at full width."""
import numpy as np
import pytest
import torch

from cikm2020_dmt_amd import ops
from cikm2020_dmt_amd import spec as S
from cikm2020_dmt_amd.data_feed.synthetic import make_batch
from cikm2020_dmt_amd.train import Trainer

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype,B", [(torch.bfloat16, 1024), (torch.float32, 96)])
def test_sequence_streams_change_nothing_but_the_schedule(cuda, dtype, B):
    ops.set_deterministic(True)            # ordered reductions: the two schedules must then agree BIT FOR BIT
    try:
        sp = S.scaled_spec(S.e64_spec(), {"Sku": 20000, "Brand": 3000, "Shopid": 3000, "Cid3": 1200})
        ref = None
        for streams, lane in ((False, False), (True, False), (True, False), (True, True), (False, True)):
            tr = Trainer(sp, device=cuda, compute_dtype=dtype, seed=5, dropout=True)
            tr.engine.seq_streams = streams
            tr.sparse_lane = lane          # id-bound tail of the step on the index lane (Trainer.train_step)
            losses = []
            for s in range(3):
                inputs, mask, _ = make_batch(sp, B, seed=80 + s, lengths="ragged", weights="random")
                losses.append(float(tr.train_step(tr.make_batch(inputs, mask))))
            if lane and B * 50 >= ops.WGRAD320_MIN_ROWS:
                assert tr.n_deferred >= 6          # the long-row weight gradients were collected and launched after backward
            tr.opt.flush_tables()
            torch.cuda.synchronize()
            state = tr.store.state_dict()
            if ref is None:
                ref = (losses, state)
                continue
            assert losses == ref[0]
            for k in state:
                assert np.array_equal(state[k].view(np.uint32), ref[1][k].view(np.uint32)), k
    finally:
        ops.set_deterministic(False)


def test_sparse_lane_with_deferred_weight_gradients_default_mode(cuda):
    """Default mode, bf16: the long-row weight gradients collected during backward and launched afterwards, the embedding-gradient tail
    and the sparse Adam on the index lane -- three steps end in the same state as the plain schedule up to atomic-order rounding."""
    sp = S.scaled_spec(S.e64_spec(), {"Sku": 20000, "Brand": 3000, "Shopid": 3000, "Cid3": 1200})
    states = []
    for lane in (False, False, True):
        tr = Trainer(sp, device=cuda, compute_dtype=torch.bfloat16, seed=5, dropout=True)
        tr.sparse_lane = lane
        for s in range(3):
            inputs, mask, _ = make_batch(sp, 2048, seed=80 + s, lengths="full")     # 2048 x 50 rows: the wgrad320 path, deferred
            tr.train_step(tr.make_batch(inputs, mask))
        assert tr.engine.step_state.deferred is None and tr.engine._pending_sparse is None
        tr.opt.flush_tables()
        torch.cuda.synchronize()
        states.append(tr.store.state_dict())
    # fp32 atomics make two runs differ, and the first Adam steps turn the sign of a rounding-level gradient into a full +-lr move; a
    # gradient term that went missing would leave its parameter where it was (moves of ~3e-3 missing: mean difference > 1e-3).  The
    # plain schedule against ITSELF is the yardstick for the total; per tensor the bound is a fraction of the three steps' move
    # (kernels that overlap reorder their atomics, so a tensor that is reproducible when run alone need not be beside the other lane).
    tot_noise = tot_lane = 0.0
    for k in states[0]:
        a, b, c = (st[k].astype(np.float64) for st in states)
        tot_noise += np.abs(a - b).mean()
        tot_lane += np.abs(a - c).mean()
        assert np.abs(a - c).mean() <= 1.5e-3, (k, np.abs(a - c).mean())     # (tensors whose true gradient is 0, e.g. the key bias, move by +-lr on noise alone)
        assert np.abs(a - c).max() <= 6.5e-3, k           # three steps at lr 1e-3: +-2 lr per step at most
    assert tot_lane <= 8.0 * tot_noise + 1e-4, (tot_lane, tot_noise)


def test_sequence_streams_full_size_default_mode(cuda):
    """Default (atomics) mode at the benchmark size: gradients of one forward/backward agree to rounding, five times over."""
    sp = S.e64_spec()
    inputs, mask, label = make_batch(sp, 4096, seed=3, lengths="full")
    res = []
    for streams in (False, True):
        tr = Trainer(sp, device=cuda, compute_dtype=torch.bfloat16, seed=2, dropout=True)
        tr.engine.seq_streams = streams
        b = tr.make_batch(inputs, mask, label)
        outs = []
        for _ in range(5 if streams else 1):
            b._prep = None
            loss = tr.forward_backward(b)
            torch.cuda.synchronize()
            outs.append((float(loss), tr.store.grads.clone(), tr.engine.sparse[2][: int(tr.engine.sparse[1].item())].clone()))
        res.append(outs)
    l0, g0, r0 = res[0][0]
    for (l1, g1, r1) in res[1]:
        assert abs(l1 - l0) < 1e-5 * abs(l0) + 1e-6
        assert ((g1 - g0).norm() / g0.norm()).item() < 2e-3
        assert (g1 - g0).abs().max().item() < 2e-2 * g0.abs().max().item()
        assert ((r1 - r0).norm() / r0.norm()).item() < 2e-3


def _lanes_worker(q, rccl):
    import os
    import torch
    from cikm2020_dmt_amd import spec as S
    from cikm2020_dmt_amd import streams
    from cikm2020_dmt_amd.train import Trainer
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    if rccl:
        import torch.distributed as dist
        os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", "29577"
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)      # (communicator set-up uses streams of its own)
    sp = S.scaled_spec(S.e64_spec(), {"Sku": 2000, "Brand": 300, "Shopid": 300, "Cid3": 120})
    Trainer(sp, device=dev, compute_dtype=torch.bfloat16, seed=1, force_dp=rccl)
    ln = streams.lanes(dev)
    groups = streams.queue_groups([torch.cuda.default_stream(dev), ln["index"], ln["seq"][0], ln["seq"][1]], ["compute", "index", "seq1", "seq2"])
    q.put((groups, ln["distinct"]))
    if rccl:
        dist.destroy_process_group()


@pytest.mark.parametrize("rccl", [False, True])
def test_lanes_are_bound_to_four_hardware_queues(cuda, rccl):
    """streams.py: the compute / index / sequence-1 / sequence-2 streams of a process each sit on their own hardware queue -- also
    after an RCCL communicator was set up first --, measured with the head-of-line probe."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p_ = ctx.Process(target=_lanes_worker, args=(q, rccl))
    p_.start()
    groups, distinct = q.get(timeout=300)
    p_.join(60)
    assert p_.exitcode == 0
    assert distinct == 4 and len(groups) == 4, (groups, distinct)

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
        for streams in (False, True, True, True):
            tr = Trainer(sp, device=cuda, compute_dtype=dtype, seed=5, dropout=True)
            tr.engine.seq_streams = streams
            losses = []
            for s in range(3):
                inputs, mask, _ = make_batch(sp, B, seed=80 + s, lengths="ragged", weights="random")
                losses.append(float(tr.train_step(tr.make_batch(inputs, mask))))
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

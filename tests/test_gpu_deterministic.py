"""Deterministic mode (ops.set_deterministic): every reduction whose partial sums normally meet in fp32 atomics
takes a fixed-order form.  Two runs of the same train steps -- at a size where hot rows span many 64-entry chunks and the weight
gradients are split over many workgroups in the default mode -- must then agree bit for bit; and the mode changes nothing but the
order of summation (default-mode results agree to fp32 rounding)."""
import numpy as np
import pytest
import torch

from cikm2020_dmt_amd import _lib as L
from cikm2020_dmt_amd import ops
from cikm2020_dmt_amd import spec as S
from cikm2020_dmt_amd.data_feed.synthetic import make_batch
from cikm2020_dmt_amd.train import Trainer

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _off_afterwards():
    yield
    ops.set_deterministic(False)


def _run(cuda, dtype, det, steps=3, B=1024):
    ops.set_deterministic(det)
    sp = S.scaled_spec(S.e64_spec(), {"Sku": 20000, "Brand": 3000, "Shopid": 3000, "Cid3": 1200})       # small vocabularies: hot rows, long runs
    tr = Trainer(sp, device=cuda, compute_dtype=dtype, seed=3, dropout=True)
    losses = []
    for s in range(steps):
        inputs, mask, _ = make_batch(sp, B, seed=50 + s, lengths="ragged", weights="random")
        losses.append(float(tr.train_step(tr.make_batch(inputs, mask))))
    tr.opt.flush_tables()
    torch.cuda.synchronize()
    return losses, tr.store.state_dict()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_deterministic_mode_is_bit_reproducible(cuda, dtype):
    l1, s1 = _run(cuda, dtype, True)
    l2, s2 = _run(cuda, dtype, True)
    assert l1 == l2
    for k in s1:
        assert np.array_equal(s1[k].view(np.uint32), s2[k].view(np.uint32)), k
    # against the default mode: same sums in another order
    l0, s0 = _run(cuda, dtype, False)
    # (bf16: the fused heads kernels sum their 1-wide layers' gradients with atomics and hand over to the layer path in deterministic
    #  mode -- another rounding of the same numbers; losses of ~10 after three steps)
    assert np.abs(np.array(l0) - np.array(l1)).max() < (1e-4 if dtype == torch.float32 else 6e-2)
    worst = max(float(np.abs(s0[k] - s1[k]).max()) for k in s1)
    # three Adam steps at lr 1e-3 bound any element's movement by 3e-3 each way: an element whose true gradient is 0 (the key bias of
    # an attention block: the softmax does not see it) moves on rounding noise alone and can end 6e-3 apart
    assert worst < 6.5e-3, worst


def test_the_library_keeps_no_mode_and_the_ops_layer_avoids_the_atomic_kernels(cuda):
    """The ordered form of a reduction is chosen per call (workspace argument / `ordered` flag): the library has no switch.  Under
    ops.set_deterministic(True) a train step takes dmt_wgrad320's ORDERED form (partial blocks through a workspace, added in split
    order) and stays away from dmt_heads_bwd (fp32 atomics by construction)."""
    lib = L.load()
    assert not hasattr(lib, "dmt_set_deterministic")
    # a too-small workspace for the ordered form is an argument error, not a silent fall back to atomics
    keys = torch.arange(100, dtype=torch.int32, device=cuda)
    seg = torch.arange(100, dtype=torch.int32, device=cuda)
    rows = torch.randn((100, 64), device=cuda)
    out = torch.zeros((100, 64), device=cuda)
    ws = torch.empty(16, dtype=torch.uint8, device=cuda)
    assert lib.dmt_rows_reduce(ops.p(keys), ops.p(keys), ops.p(seg), 100, 1000, ops.p(rows), ops.p(out), 64, ops.p(ws), 16, ops.stream_ptr()) == -1
    assert b"workspace" in lib.dmt_last_error()
    # ordered and atomic column sums agree to rounding; the ordered one is bit-reproducible
    x = torch.randn((5000, 333), device=cuda)
    outs = []
    for ordered in (1, 1, 0):
        o = torch.zeros(333, device=cuda)
        L.call("dmt_colsum", L.DMT_F32, 5000, 333, ops.p(x), 333, 1.0, ops.p(o), ordered, ops.stream_ptr())
        outs.append(o)
    assert torch.equal(outs[0], outs[1]) and torch.allclose(outs[0], outs[2], rtol=1e-4, atol=1e-3)
    sp = S.scaled_spec(S.e64_spec(), {"Sku": 2000, "Brand": 300, "Shopid": 300, "Cid3": 120})
    inputs, mask, _ = make_batch(sp, 400, seed=1, lengths="full")
    for det in (True, False):
        ops.set_deterministic(det)
        tr = Trainer(sp, device=cuda, compute_dtype=torch.bfloat16, seed=3, dropout=True)
        with L.route_trace() as rt:
            tr.train_step(tr.make_batch(inputs, mask))
            torch.cuda.synchronize()
        ordered, heads = rt.counts.get("dmt_wgrad320(ordered reduce)", 0), rt.counts.get("dmt_heads_bwd", 0)
        assert rt.counts.get("dmt_wgrad320", 0) > 0, rt.counts
        assert (ordered > 0 and heads == 0) if det else (ordered == 0 and heads > 0), rt.counts

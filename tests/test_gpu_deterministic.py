"""Deterministic mode: with dmt_set_deterministic(1) every reduction whose partial sums normally meet in fp32 atomics
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
    assert L.load().dmt_get_deterministic() == 0
    l1, s1 = _run(cuda, dtype, True)
    assert L.load().dmt_get_deterministic() == 1
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


def test_atomic_kernels_refuse_in_deterministic_mode(cuda):
    import ctypes as C
    ops.set_deterministic(True)
    lib = L.load()
    x = torch.randn((32768, 320), device=cuda).to(torch.bfloat16)
    dz = torch.randn((32768, 256), device=cuda).to(torch.bfloat16)
    gw = torch.zeros((320, 256), device=cuda)
    d = L.WgradDesc()
    d.A, d.ld_a, d.a_cols, d.B, d.ld_b, d.M, d.N, d.C, d.ldc = x.data_ptr(), 320, 320, dz.data_ptr(), 256, 32768, 256, gw.data_ptr(), 256
    assert lib.dmt_wgrad320(C.byref(d), ops.stream_ptr()) == -3
    with pytest.raises(L.DmtError, match="split_k"):
        ops.gemm(x, 1, 320, dz, 256, 1, 320, 256, 32768, gw, 256, split_k=4, accumulate=True)
    # the segmented reductions need their workspace
    keys = torch.arange(100, dtype=torch.int32, device=cuda)
    seg = torch.arange(100, dtype=torch.int32, device=cuda)
    rows = torch.randn((100, 64), device=cuda)
    out = torch.zeros((100, 64), device=cuda)
    assert lib.dmt_rows_reduce(ops.p(keys), ops.p(keys), ops.p(seg), 100, 1000, ops.p(rows), ops.p(out), 64, None, 0, ops.stream_ptr()) == -1
    assert b"workspace" in lib.dmt_last_error()
    torch.cuda.synchronize()

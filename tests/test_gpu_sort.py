"""dmt_sort_pairs / dmt_segment_heads (csrc/dmt_sort.hip, the library's own radix sort of the index plane) against numpy's stable argsort and
np.unique: bit-exact, every tile / chunk / digit-width edge, Zipf-like pile-ups (every lane of a wavefront on one digit), the invalid key."""
import ctypes as C

import numpy as np
import pytest
import torch

from cikm2020_dmt_amd import _lib as L
from cikm2020_dmt_amd import ops

pytestmark = pytest.mark.gpu


def _sort(keys, vals, end_bit, dev):
    n = keys.size
    k = torch.as_tensor(keys.view(np.int32)).to(dev)
    v = torch.as_tensor(vals.view(np.int32)).to(dev) if vals is not None else None
    ks, vs = torch.empty_like(k), torch.empty_like(k)
    need = C.c_uint64(0)
    L.call("dmt_sort_pairs", ops.p(k), ops.p(ks), ops.p(v), ops.p(vs), n, end_bit, None, C.byref(need), ops.stream_ptr())
    ws = torch.empty((max(int(need.value), 16),), dtype=torch.uint8, device=dev)
    have = C.c_uint64(ws.numel())
    L.call("dmt_sort_pairs", ops.p(k), ops.p(ks), ops.p(v), ops.p(vs), n, end_bit, ops.p(ws), C.byref(have), ops.stream_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(k.cpu().numpy().view(np.uint32), keys)          # inputs untouched
    return ks, vs


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 1023, 4095, 4096, 4097, 8192 + 5, 4096 * 33 + 17, 4096 * 70, 1_000_003])
@pytest.mark.parametrize("end_bit,law", [(1, "uniform"), (7, "uniform"), (8, "zipf"), (9, "uniform"), (16, "zipf"), (23, "zipf"), (23, "uniform"),
                                         (27, "uniform"), (32, "uniform"), (23, "constant")])
def test_sort_pairs_is_numpys_stable_sort(cuda, n, end_bit, law):
    rng = np.random.default_rng(n * 37 + end_bit)
    hi = (1 << end_bit) - 1
    if law == "uniform":
        keys = rng.integers(0, hi + 1, n, dtype=np.uint64).astype(np.uint32)
    elif law == "zipf":
        keys = np.minimum(rng.zipf(1.05, n), hi).astype(np.uint32)            # half of the entries on a handful of keys
    else:
        keys = np.full(n, min(hi, 4242), dtype=np.uint32)
    for given in (False, True):
        vals = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32) if given else None
        ks, vs = _sort(keys, vals, end_bit, cuda)
        order = np.argsort(keys, kind="stable")
        assert np.array_equal(ks.cpu().numpy().view(np.uint32), keys[order])
        want_v = vals[order] if given else order.astype(np.uint32)
        assert np.array_equal(vs.cpu().numpy().view(np.uint32), want_v)


def test_sort_ignores_bits_above_end_bit(cuda):
    """Only bits [0, end_bit) order the pairs (the callers' keys are < 2^end_bit; stability covers the rest)."""
    rng = np.random.default_rng(5)
    n = 50_000
    keys = rng.integers(0, 1 << 20, n, dtype=np.uint64).astype(np.uint32)
    ks, vs = _sort(keys, None, 12, cuda)
    order = np.argsort(keys & 0xFFF, kind="stable")
    assert np.array_equal(ks.cpu().numpy().view(np.uint32), keys[order])
    assert np.array_equal(vs.cpu().numpy().view(np.uint32), order.astype(np.uint32))


@pytest.mark.parametrize("n", [1, 15, 16, 17, 4095, 4096, 4097, 4096 * 1025 + 3, 300_001])
@pytest.mark.parametrize("distinct", [1, 3, 1000, 10 ** 9])
def test_segment_heads_matches_numpy_unique(cuda, n, distinct):
    rng = np.random.default_rng(n + distinct)
    invalid = 1 << 22
    keys = np.sort(rng.integers(0, min(distinct, invalid), n, dtype=np.int64)).astype(np.uint32)
    n_inv = int(rng.integers(0, max(1, n // 3)))
    for tail in (0, n_inv):
        k = keys.copy()
        if tail:
            k[n - tail:] = invalid                     # padding entries carry the invalid key and sort last
        kd = torch.as_tensor(k.view(np.int32)).to(cuda)
        seg = torch.empty((n,), dtype=torch.int32, device=cuda)
        uniq = torch.full((n,), -1, dtype=torch.int32, device=cuda)
        nu = torch.full((1,), -7, dtype=torch.int32, device=cuda)
        need = C.c_uint64(0)
        L.call("dmt_segment_heads", ops.p(kd), n, invalid, ops.p(seg), ops.p(uniq), ops.p(nu), None, C.byref(need), ops.stream_ptr())
        ws = torch.empty((max(int(need.value), 16),), dtype=torch.uint8, device=cuda)
        have = C.c_uint64(ws.numel())
        L.call("dmt_segment_heads", ops.p(kd), n, invalid, ops.p(seg), ops.p(uniq), ops.p(nu), ops.p(ws), C.byref(have), ops.stream_ptr())
        torch.cuda.synchronize()
        u, inv = np.unique(k, return_inverse=True)
        assert np.array_equal(seg.cpu().numpy(), inv.astype(np.int32))
        assert np.array_equal(uniq.cpu().numpy()[: u.size].view(np.uint32), u)
        assert int(nu.item()) == int((u < invalid).sum())


def test_no_vendor_sort_symbols_in_the_library():
    """The index plane's sort is the library's own: no rocPRIM kernel is linked into libdmt_hip.so."""
    import os
    import subprocess
    so = os.path.join(os.path.dirname(L.__file__), "csrc", "libdmt_hip.so")
    out = subprocess.run(["strings", so], capture_output=True, text=True).stdout
    assert "rocprim" not in out


def test_sort_soak_against_torch_stable_sort(cuda):
    """300 sorts of fresh data at the step's size class (0.5 - 6 M pairs, Zipf and uniform keys, 20 - 27 key bits) against torch's stable sort on the
    device: the ranking's LDS read-then-write ordering inside a wavefront and the integer atomics of the chunk totals must hold every time."""
    g = torch.Generator(device=cuda).manual_seed(7)
    need = C.c_uint64(0)
    for it in range(300):
        n = int(torch.randint(500_000, 6_000_000, (1,), generator=g, device=cuda).item()) if it % 3 else int(torch.randint(1, 70_000, (1,), generator=g, device=cuda).item())
        bits = 20 + it % 8
        if it % 2:
            u = torch.rand(n, generator=g, device=cuda)
            keys = ((1.0 / (u + 1e-7)) ** 1.2).clamp_(max=2.0 ** 30).to(torch.int64).clamp_(max=(1 << bits) - 1)      # heavy head: most keys on a few rows
        else:
            keys = torch.randint(0, 1 << bits, (n,), generator=g, device=cuda, dtype=torch.int64)
        k32 = keys.to(torch.int32)
        ks, vs = torch.empty_like(k32), torch.empty_like(k32)
        L.call("dmt_sort_pairs", ops.p(k32), ops.p(ks), None, ops.p(vs), n, bits, None, C.byref(need), ops.stream_ptr())
        ws = torch.empty((max(int(need.value), 16),), dtype=torch.uint8, device=cuda)
        ws.random_(0, 255)                                   # (the workspace arrives dirty: nothing may depend on its content)
        have = C.c_uint64(ws.numel())
        L.call("dmt_sort_pairs", ops.p(k32), ops.p(ks), None, ops.p(vs), n, bits, ops.p(ws), C.byref(have), ops.stream_ptr())
        want_k, want_i = torch.sort(keys, stable=True)
        assert torch.equal(ks.to(torch.int64), want_k), (it, n, bits)
        assert torch.equal(vs.to(torch.int64), want_i), (it, n, bits)

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    from cikm2020_dmt_amd import _lib
    _lib.load()   # fail loudly if the HIP library is missing on a GPU box
    return torch.device("cuda:0")


@pytest.fixture(autouse=True)
def _nan_filled_free_memory(request):
    """GPU tests start from free device memory filled with NaNs (the caching allocator's free blocks, a few hundred MB plus one block of
    each common size): a kernel that reads memory nobody wrote cannot pass by finding zeros there.  DMT_TEST_POLLUTE=off disables."""
    if request.node.get_closest_marker("gpu") is None or os.environ.get("DMT_TEST_POLLUTE", "nan") == "off":
        yield
        return
    import torch
    if torch.cuda.is_available():
        fill = float(os.environ.get("DMT_TEST_POLLUTE", "nan"))
        junk = [torch.full((1 << 25,), fill, device="cuda:0") for _ in range(2)]
        junk += [torch.full((n,), fill, device="cuda:0") for n in (7, 64, 300, 4096, 20000, 70000, 1 << 20, 1 << 23) for _ in range(4)]
        torch.cuda.synchronize()
        del junk
    yield

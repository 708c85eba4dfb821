"""Host-side cost of one train step: how long the Python / ctypes / autograd side takes to ENQUEUE a step when nothing holds it back
(queues empty after a synchronize, a burst of n steps, n small enough that no queue fills), against the GPU time of the same steps.

    python scripts/host_time.py [burst] [full|ragged]
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cikm2020_dmt_amd import spec as S                      # noqa: E402
from cikm2020_dmt_amd.data_feed.synthetic import make_batch  # noqa: E402
from cikm2020_dmt_amd.train import Trainer                  # noqa: E402


def main(burst, lengths="full"):
    sp = S.e64_spec()
    tr = Trainer(sp, device="cuda:0", compute_dtype=torch.bfloat16, seed=1234, dropout=True)
    batches = [tr.make_batch(*make_batch(sp, 4096, seed=7 + i, lengths=lengths, law="zipf")) for i in range(4)]

    def step(i):
        b, nxt = batches[i % 4], batches[(i + 1) % 4]
        nxt._prep = None
        return tr.train_step(b, prefetch=nxt)
    for i in range(12):
        step(i)
    torch.cuda.synchronize()
    for rep in range(4):
        t0 = time.perf_counter()
        for i in range(burst):
            step(i)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("burst of %d steps: host enqueue %.3f ms/step, until the GPU is done %.3f ms/step" % (burst, (t1 - t0) / burst * 1e3, (t2 - t0) / burst * 1e3))
    import cProfile
    import pstats
    pr = cProfile.Profile()
    # (backward in the calling thread: the autograd engine's device thread is invisible to cProfile)
    with torch.autograd.set_multithreading_enabled(False):
        step(0)
        pr.enable()
        for i in range(burst):
            step(i)
        pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(70)
    pstats.Stats(pr).sort_stats("tottime").print_stats(45)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 3, sys.argv[2] if len(sys.argv) > 2 else "full")

"""The HIP streams of a training process, bound to hardware queues ON PURPOSE.

Measured on MI355X / ROCm 7.2 (scripts/stream_queues.py, scripts/stream_queues2.py): the runtime gives a process four hardware
queues for its normal-priority streams and binds a stream to one of them at the stream's FIRST USE -- the first four streams used
get a queue each, later ones the least-loaded queue (5th with the 4th, 6th with the 3rd, 7th with the 2nd, 8th with the 1st).
Streams that share a queue are not serialised kernel by kernel, but a packet waits until every packet issued to that queue before
it has been dispatched: work issued behind a long dependent chain of another stream starts when that chain is nearly done.  A fifth
queue (GPU_MAX_HW_QUEUES > 4, or a high-priority stream) made the step 40 % SLOWER.  So the step's concurrent lanes are laid out as

    queue A   compute stream (torch's default stream)      + the gradient collectives' RCCL stream
    queue B   index plane (id sort, exchange plan)          + the id exchange's RCCL stream (second communicator)
    queue C   behaviour sequence 1
    queue D   behaviour sequence 2
    (two spare streams are bound to D and C only to advance the runtime's round-robin to B and A for the RCCL streams)

How a later stream is placed depends on what the process used before (RCCL's communicator set-up alone uses several internal
streams), so lanes() does not assume the order: it takes pool streams one by one, uses each once, PROBES which of the already chosen
lanes it shares a queue with (queue_groups: ~5 ms per pair) and keeps the first three that sit on queues of their own next to the
compute stream.  One set per device and process, shared by every Trainer / engine (tests/test_gpu_streams.py checks the result).
The RCCL streams cannot be chosen; warm_communicators() only fixes WHEN they are bound (index-plane communicator first).
"""
from __future__ import annotations

import torch

_lanes = {}
_warm = {"world": None}


def _touch(stream, device):
    with torch.cuda.stream(stream):
        torch.empty(8, device=device).zero_()       # any kernel launch binds the stream to its queue


def _shares_queue(a, b, device, cycles=2_000_000):
    """Head-of-line probe: two dependent sleeps on a, then one on b: b's ends after ONE sleep on another queue, after TWO on a's."""
    def run(x, y):
        torch.cuda.synchronize(device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(y):
            e0.record()
        with torch.cuda.stream(x):
            torch.cuda._sleep(cycles)
            torch.cuda._sleep(cycles)
        with torch.cuda.stream(y):
            torch.cuda._sleep(cycles)
            e1.record()
        torch.cuda.synchronize(device)
        return e0.elapsed_time(e1)
    # the yardstick (one sleep) is taken before AND after the pair, smallest wins: clocks still ramping up during the first
    # measurement would otherwise make a shared queue look like two
    one = run(a, a) / 3
    t = run(a, b)
    one = min(one, run(a, a) / 3)
    return t > 1.6 * one


def lanes(device) -> dict:
    """{'index': stream, 'seq': [stream, stream], 'spare': [...], 'distinct': n} of `device`: streams on hardware queues of their own
    (as far as the runtime has them), chosen by probing on the first call."""
    device = torch.device(device)
    key = device.index if device.index is not None else torch.cuda.current_device()
    got = _lanes.get(key)
    if got is None:
        dev = torch.device("cuda", key)
        main = torch.cuda.default_stream(dev)
        _touch(main, dev)
        chosen, spare = [], []
        for _ in range(12):
            if len(chosen) == 3:
                break
            s = torch.cuda.Stream(dev)
            _touch(s, dev)
            if any(_shares_queue(c, s, dev) for c in [main] + chosen):
                spare.append(s)
            else:
                chosen.append(s)
        distinct = 1 + len(chosen)
        while len(chosen) < 3:                       # fewer queues than lanes: share (correct, less overlap)
            chosen.append(spare.pop() if spare else torch.cuda.Stream(dev))
        torch.cuda.synchronize(dev)
        got = _lanes[key] = dict(index=chosen[0], seq=[chosen[1], chosen[2]], spare=spare, distinct=distinct)
    return got


def warm_communicators(device):
    """First collective of the index-plane communicator, then of the default one: their RCCL streams are the 7th and 8th streams the
    process uses -> queues B and A.  Collective: every rank builds its Trainer at the same point.  No-op without RCCL."""
    import torch.distributed as dist
    from . import parallel
    if not (dist.is_available() and dist.is_initialized()) or dist.get_backend() != "nccl":
        return
    world_pg = dist.distributed_c10d._get_default_group()
    if _warm["world"] is world_pg:
        return
    _warm["world"] = world_pg
    lanes(device)
    t = torch.zeros(1, device=device)
    dist.all_reduce(t, group=parallel.index_group())
    dist.all_reduce(t)
    torch.cuda.synchronize(device)


def queue_groups(streams, names, device=None):
    """Which of `streams` share a hardware queue?  -> list of groups of names."""
    device = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    n = len(streams)
    groups = list(range(n))
    for i in range(n):
        for j in range(i + 1, n):
            if _shares_queue(streams[i], streams[j], device):
                gi, gj = groups[i], groups[j]
                groups = [gi if g == gj else g for g in groups]
    out = {}
    for nm, g in zip(names, groups):
        out.setdefault(g, []).append(nm)
    return list(out.values())

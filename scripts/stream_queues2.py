"""Does the runtime bind a stream to a hardware queue at creation (pool order) or at first use?  Six pool streams, FIRST USED IN
REVERSE ORDER, then the head-of-line probe of scripts/stream_queues.py.  Also prints torch's pool index of each stream."""
import torch

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
ss = [torch.cuda.Stream(dev) for _ in range(6)]
print("pool index:", [(s.stream_id >> 5) & 31 for s in ss], "type:", [(s.stream_id >> 1) & 15 for s in ss])
for s in reversed(ss):
    with torch.cuda.stream(s):
        torch.zeros(4, device=dev).add_(1)
torch.cuda.synchronize()
streams = [torch.cuda.default_stream(dev)] + ss
CY = 4_000_000


def pair(a, b):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(b):
        e0.record()
    with torch.cuda.stream(a):
        torch.cuda._sleep(CY)
        torch.cuda._sleep(CY)
    with torch.cuda.stream(b):
        torch.cuda._sleep(CY)
        e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3


one = pair(ss[0], ss[0]) / 3
n = len(streams)
groups = list(range(n))
for i in range(n):
    for j in range(i + 1, n):
        if pair(streams[i], streams[j]) > 1.6 * one:
            gi, gj = groups[i], groups[j]
            groups = [gi if g == gj else g for g in groups]
names = ["null"] + ["s%d" % k for k in range(6)]
out = {}
for nm, g in zip(names, groups):
    out.setdefault(g, []).append(nm)
print(list(out.values()))

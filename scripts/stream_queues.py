"""Which HIP streams of one process share a hardware queue?  Two sleeps on streams of the same queue serialise.
Prints, for the null stream (index -1) and 10 created streams, the groups that serialise with each other."""
import time
import torch

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
x = torch.zeros(1, device=dev)
streams = [torch.cuda.default_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(10)]
hi = torch.cuda.Stream(dev, priority=-1)
streams.append(hi)
CY = 20_000_000     # ~8 ms at 2.4 GHz


def pair(a, b):
    """a: two dependent sleeps; b: one sleep issued after them.  -> when b's sleep finished, in sleeps.  Streams that share a hardware
    queue: b's packet sits behind a's second one, which waits for a's first (head-of-line blocking) -> 2; otherwise 1."""
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


one = pair(streams[1], streams[1]) / 3
print("one sleep %.2f ms" % (one * 1e3))
n = len(streams)
groups = list(range(n))
for i in range(n):
    for j in range(i + 1, n):
        t = pair(streams[i], streams[j])
        if t > 1.6 * one:
            gi, gj = groups[i], groups[j]
            groups = [gi if g == gj else g for g in groups]
names = ["null"] + ["s%d" % k for k in range(10)] + ["hi"]
out = {}
for nm, g in zip(names, groups):
    out.setdefault(g, []).append(nm)
print(list(out.values()))

"""Per-kernel summary of a rocprofv3 --kernel-trace csv of `python bench.py ...`, split into the two phases of a bench run:
the TIMED steps (three sequence lanes in flight: kernels of different lanes overlap) and the EXCLUSIVE steps bench.py runs right
after them with the lanes serialised (the durations its `roofline` quotes).  Steps are delimited by embgrad_keys_kernel (one per step).
usage: python scripts/kernel_summary.py <kernel_trace.csv> [exclusive steps at the end, default 4] [warmup+steps, default: all but the last 5]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n_x = int(sys.argv[2]) if len(sys.argv) > 2 else 4
n_t = int(sys.argv[3]) if len(sys.argv) > 3 else None
# a kernel belongs to the step whose first forward kernel (gather_group_kernel, first of a burst) precedes it
starts = []
last = -10**18
for r in rows:
    if "gather_group_kernel" in r["Kernel_Name"]:
        t = int(r["Start_Timestamp"])
        if t - last > 2_000_000:
            starts.append(t)
        last = t
steps = len(starts)
if n_t is None:
    n_t = steps - n_x - 1
print("steps in the trace: %d (last %d exclusive; first %d = warm-up + timed)" % (steps, n_x + 1, n_t))
bounds_t = (starts[2], starts[n_t])                      # skip the first two warm-up steps
bounds_x = (starts[steps - n_x], int(rows[-1]["End_Timestamp"]) + 1) if steps > n_t else None
acc = {"t": defaultdict(lambda: [0, 0]), "x": defaultdict(lambda: [0, 0])}
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nm = r["Kernel_Name"].replace("(anonymous namespace)::", "")
    if bounds_t[0] <= s < bounds_t[1]:
        a = acc["t"][nm]; a[0] += 1; a[1] += e - s
    elif bounds_x and bounds_x[0] <= s < bounds_x[1]:
        a = acc["x"][nm]; a[0] += 1; a[1] += e - s
nt, nx = n_t - 2, n_x
tot_t = sum(v[1] for v in acc["t"].values()) / nt / 1e6
tot_x = sum(v[1] for v in acc["x"].values()) / max(nx, 1) / 1e6
span_t = (bounds_t[1] - bounds_t[0]) / nt / 1e6
print("timed steps: %.3f ms wall per step, %.3f ms of kernel time per step (lanes overlap); exclusive steps: %.3f ms of kernel time per step" % (span_t, tot_t, tot_x))
print("launches per step: %.1f (timed), %.1f (exclusive)" % (sum(v[0] for v in acc["t"].values()) / nt, sum(v[0] for v in acc["x"].values()) / max(nx, 1)))
print("%-88s %7s %10s %10s %10s" % ("kernel", "calls", "ms/step", "avg us", "avg us"))
print("%-88s %7s %10s %10s %10s" % ("", "/step", "(timed)", "(timed)", "(exclusive)"))
for nm, (c, d) in sorted(acc["t"].items(), key=lambda kv: -kv[1][1])[:45]:
    cx, dx = acc["x"].get(nm, [0, 0])
    print("%-88s %7.1f %10.3f %10.1f %10s" % (nm[:88], c / nt, d / nt / 1e6, d / c / 1e3, ("%.1f" % (dx / cx / 1e3)) if cx else "-"))

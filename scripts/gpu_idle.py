"""How much of a steady-state step is the GPU idle (no kernel running) / running exactly one kernel?  From a rocprofv3 --kernel-trace csv.
usage: python scripts/gpu_idle.py <kernel_trace.csv> [steps from the end to skip, default 6] [steps to analyse, default 20]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 6
nst = int(sys.argv[3]) if len(sys.argv) > 3 else 20
marks = [int(r["Start_Timestamp"]) for r in rows if "adam_catchup_kernel" in r["Kernel_Name"]]
a, b = marks[-skip - nst], marks[-skip]
ev = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e <= a or s >= b: continue
    ev.append((max(s, a), 1)); ev.append((min(e, b), -1))
ev.sort()
lvl, last, hist = 0, a, {}
gaps = []
for t, d in ev:
    hist[lvl] = hist.get(lvl, 0) + (t - last)
    if lvl == 0 and t - last > 0: gaps.append((t - last, last - a))
    last = t; lvl += d
hist[lvl] = hist.get(lvl, 0) + (b - last)
tot = b - a
print("window %.3f ms = %d steps of %.3f ms" % (tot / 1e6, nst, tot / 1e6 / nst))
for k in sorted(hist): print("  %d kernels running: %5.1f %% (%.3f ms/step)" % (k, 100.0 * hist[k] / tot, hist[k] / 1e6 / nst))
gaps.sort(reverse=True)
print("idle gaps per step: %.1f of > 2 us, largest (us): %s" % (sum(1 for g, _ in gaps if g > 2000) / nst, [round(g / 1e3, 1) for g, _ in gaps[:8]]))

"""Where one training step leaves the chip idle or nearly idle, from a rocprofv3 --kernel-trace csv:
   (1) the gaps with NO kernel running (start, length, the kernels before / after), (2) the time covered by exactly one kernel whose grid
   cannot fill 256 CUs (small launches), per kernel name.
usage: python scripts/step_gaps.py <kernel_trace.csv> [step index from the end, default 20] [min gap us, default 3]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 20
min_gap = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
ks = [i for i, r in enumerate(rows) if "embgrad_reduce" in r["Kernel_Name"]]
a, b = ks[-back - 1], ks[-back]
seg = rows[a:b]
t0 = int(seg[0]["Start_Timestamp"])
span = (int(rows[b]["Start_Timestamp"]) - t0) / 1e3
iv = [((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r) for r in seg]


def short(r):
    return r["Kernel_Name"][:48].replace("(anonymous namespace)::", "")


def wgs(r):
    try:
        return (int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])) // max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))
    except Exception:
        return -1


print("step (embgrad_reduce to embgrad_reduce): %.1f us, %d kernels, sum of kernel durations %.1f us" % (span, len(seg), sum(e - s for s, e, _ in iv)))
# sweep
ev = sorted([(s, 1, i) for i, (s, e, _) in enumerate(iv)] + [(e, -1, i) for i, (s, e, _) in enumerate(iv)])
active, last_t, idle, one = set(), 0.0, 0.0, {}
gaps = []
last_end_kernel = None
for t, d, i in ev:
    if t > last_t:
        if not active:
            idle += t - last_t
            if t - last_t >= min_gap:
                gaps.append((last_t, t - last_t, last_end_kernel, None))
        elif len(active) == 1:
            (j,) = tuple(active)
            r = iv[j][2]
            if 0 <= wgs(r) < 256:
                one[short(r)] = one.get(short(r), 0.0) + (t - last_t)
    if d == 1:
        if gaps and gaps[-1][3] is None and not active:
            gaps[-1] = gaps[-1][:3] + (short(iv[i][2]),)
        active.add(i)
    else:
        active.discard(i)
        last_end_kernel = short(iv[i][2])
    last_t = t
print("no kernel running: %.1f us (%.1f %% of the step); gaps >= %.0f us:" % (idle, 100 * idle / span, min_gap))
for s, l, k0, k1 in gaps:
    print("  %9.1f  %7.1f us   after %-50s before %s" % (s, l, k0, k1))
print("ONE kernel with < 256 workgroups alone on the chip: %.1f us" % sum(one.values()))
for k, v in sorted(one.items(), key=lambda x: -x[1])[:25]:
    print("  %8.1f us  %s" % (v, k))

"""Timeline of ONE training step out of a rocprofv3 --kernel-trace csv: per kernel start / duration / queue.
usage: python scripts/step_timeline.py <kernel_trace.csv> [step index from the end, default 5]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ks = [i for i, r in enumerate(rows) if "embgrad_keys" in r["Kernel_Name"]]
a, b = ks[-back - 1], ks[-back]
seg = rows[a:b]
t0 = int(seg[0]["Start_Timestamp"])
print("step span ms", (int(rows[b]["Start_Timestamp"]) - t0) / 1e6, "kernels", len(seg))
for r in seg:
    s = (int(r["Start_Timestamp"]) - t0) / 1e3
    e = (int(r["End_Timestamp"]) - t0) / 1e3
    print("%8.1f %8.1f q%s %s" % (s, e - s, r.get("Queue_Id"), r["Kernel_Name"][:60].replace("(anonymous namespace)::", "")))

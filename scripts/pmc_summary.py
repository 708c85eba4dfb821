import csv, glob, collections, sys
pat = sys.argv[2] if len(sys.argv) > 2 else "gemm_kernel"
for d in sys.argv[1].split(","):
    fs = glob.glob("%s/*/*counter_collection.csv" % d)
    if not fs:
        print(d, "no counter file"); continue
    rows = list(csv.DictReader(open(fs[0])))
    agg = collections.OrderedDict()
    for r in rows:
        if pat not in r["Kernel_Name"]: continue
        key = (r["Kernel_Name"].split("(")[0][-44:], r["Grid_Size"], r["Counter_Name"])
        agg.setdefault(key, []).append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print("%-46s grid %-9s %-26s n=%d avg=%.5g" % (k[0], k[1], k[2], len(v), sum(v) / len(v)))

"""Aggregate FETCH_SIZE / WRITE_SIZE (rocprofv3 --pmc, one counter per pass) over the GEMM kernels of a bench.py run.
usage: pmc_traffic.py <fetch_dir> <write_dir> <steps_total> -> JSON on stdout (bytes per GEMM launch, gfx950 FETCH x2 correction)"""
import csv, glob, json, sys
def load(d, counter):
    f = glob.glob("%s/*/*counter_collection.csv" % d)[0]
    tot, n = 0.0, 0
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter: continue
        k = r["Kernel_Name"]
        if "gemm_glds_kernel" in k or "gemm_dw_glds_kernel" in k or "gemm_kernel<" in k:
            tot += float(r["Counter_Value"]); n += 1
    return tot, n
fetch_kb, n1 = load(sys.argv[1], "FETCH_SIZE")
write_kb, n2 = load(sys.argv[2], "WRITE_SIZE")
assert n1 == n2 and n1 > 0, (n1, n2)
# MI355X_MICROARCH.md (HBM): FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half of a wide streaming read
fetch_b = fetch_kb * 1024 * 2
write_b = write_kb * 1024
print(json.dumps({"launches": n1, "fetch_bytes_per_launch": fetch_b / n1, "write_bytes_per_launch": write_b / n1,
                  "hbm_bytes_per_launch": (fetch_b + write_b) / n1,
                  "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python bench.py --steps 3 --warmup 1 --no-cpu-baseline`; GEMM kernels only; FETCH_SIZE doubled (gfx950 half-count), WRITE_SIZE as reported"}))

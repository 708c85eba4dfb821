R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/dpm -- python $R/scripts/dp_merge_bench.py 8 > $R/gpurun_out/dpm.log 2>&1
f=$(find $R/gpurun_out/dpm -name "*kernel_trace.csv" | head -1)
python - <<PY
import csv, collections
rows=list(csv.DictReader(open("$f")))
# last 12 merge_gathered calls happen before the optimizer timing; aggregate by kernel over the whole run for the merge-related kernels
agg=collections.defaultdict(list)
for r in rows:
    n=r["Kernel_Name"]
    if any(k in n for k in ("rows_reduce","zero_rows","rocprim","head_flags","finish_heads","adam_sparse","adam_dense","iota","arange","elementwise")):
        agg[n.split("(")[0][-60:]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in sorted(agg.items(), key=lambda kv:-max(kv[1])):
    v2=sorted(v)[-12:]
    print("%-62s n=%4d  top-12 avg %8.1f us  max %8.1f" % (k, len(v), sum(v2)/len(v2), max(v)))
PY

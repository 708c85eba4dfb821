R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/at1 -- python $R/scripts/attn_one.py 5 > $R/gpurun_out/at1.log 2>&1
f=$(find $R/gpurun_out/at1 -name "*kernel_stats.csv" | head -1)
python - <<PY
import csv
for r in list(csv.DictReader(open("$f")))[:6]:
    print("%-70s calls %4s avg %8.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3))
PY

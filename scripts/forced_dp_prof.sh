R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
DMT_BENCH_FORCE_DP=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/fdp -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/fdp.log 2>&1
f=$(ls -t $R/gpurun_out/fdp/*/*kernel_stats.csv | head -1)
python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total GPU ms/step %.3f" % (tot/7e6))
base={}
for r in rows[:60]:
    print("%-86s calls %5s %8.3f ms/step avg %8.1f us" % (r["Name"].replace("(anonymous namespace)::","")[:86], r["Calls"], float(r["TotalDurationNs"])/7e6, float(r["AverageNs"])/1e3))
PY
grep "^{" $R/gpurun_out/fdp.log | cut -c1-160

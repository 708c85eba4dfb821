#!/bin/bash
# usage: scripts/kstats.sh <outdir-name> [bench args...]   (run on the GPU box, from the repo root)
R=${GRAFT_REPO_ROOT:-$(pwd)}
name=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$name -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline "$@" > $R/gpurun_out/$name.log 2>&1
f=$(find $R/gpurun_out/$name -name "*kernel_stats.csv" | head -1)
python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total GPU ms/step %.3f" % (tot/7e6))
for r in rows[:26]:
    print("%-64s calls %5s %8.3f ms/step %5.1f%% avg %8.1f us" % (r["Name"].replace("(anonymous namespace)::","")[:64], r["Calls"], float(r["TotalDurationNs"])/7e6, float(r["Percentage"]), float(r["AverageNs"])/1e3))
PY
tail -1 $R/gpurun_out/$name.log | cut -c1-160

# HBM traffic per launch of the kernel families of bench.py (Zipf and uniform ids): FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3
# --pmc passes, each under its own timeout.  usage: scripts/pmc_traffic.sh <tag>     -> gpurun_out/<tag>_traffic.json
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1
cd /tmp && export TMPDIR=/tmp
for law in zipf uniform; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/${tag}_${law}_$c -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --record-files 0 --law $law > $R/gpurun_out/${tag}_${law}_$c.log 2>&1 || echo "pass $law $c failed"
  done
done
cd $R
python - <<PY
import json, subprocess, sys
sys.path.insert(0, ".")
import bench
out = {"kernel_source_sha": bench.kernel_source_sha(), "note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over python bench.py --steps 3 --warmup 1 --no-cpu-baseline --law <law>; 9 steps per pass (1 warm-up + 3 timed + 1 + 4 with the lanes serialised); bytes per launch per kernel family; FETCH_SIZE x 1024 x 2 (gfx950 counts 64 B per 128-B request), WRITE_SIZE x 1024"}
for law in ("zipf", "uniform"):
    out[law] = json.loads(subprocess.check_output(["python", "scripts/pmc_traffic.py", "gpurun_out/${tag}_%s_FETCH_SIZE" % law, "gpurun_out/${tag}_%s_WRITE_SIZE" % law, "9"]))
json.dump(out, open("gpurun_out/${tag}_traffic.json", "w"), indent=1)
print(json.dumps(out)[:3000])
PY

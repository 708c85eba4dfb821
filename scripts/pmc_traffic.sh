# HBM traffic of the GEMM launches of bench.py: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (two TCC-derived
# counters in one pass hung on this pool), each under its own timeout.  usage: scripts/pmc_traffic.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/${tag}_$c -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${tag}_$c.log 2>&1 || echo "pass $c failed"
done
cd $R
python scripts/pmc_traffic.py gpurun_out/${tag}_FETCH_SIZE gpurun_out/${tag}_WRITE_SIZE 4 > gpurun_out/${tag}_gemm_traffic.json && cat gpurun_out/${tag}_gemm_traffic.json

R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
DMT_BENCH_FORCE_DP=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/fdps -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --shard-tables --sku-rows 100000000 > $R/gpurun_out/fdps.log 2>&1
cd $R
python scripts/kernel_summary.py $(ls -t gpurun_out/fdps/*/*_kernel_trace.csv | head -1) 2 > gpurun_out/fdps_summary.txt 2>&1
rm -f gpurun_out/fdps/*/*_kernel_trace.csv
head -75 gpurun_out/fdps_summary.txt | cut -c1-150

#!/bin/bash
# usage: scripts/ksummary.sh <outdir-name> [bench args...]   (GPU box, repo root): per-step kernel summary (timed + exclusive averages)
R=${GRAFT_REPO_ROOT:-$(pwd)}
name=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$name -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > $R/gpurun_out/$name.log 2>&1
cd $R
python scripts/kernel_summary.py $(ls -t gpurun_out/$name/*/*_kernel_trace.csv | head -1) > gpurun_out/${name}_summary.txt 2>&1
rm -f gpurun_out/$name/*/*_kernel_trace.csv
head -60 gpurun_out/${name}_summary.txt | cut -c1-150

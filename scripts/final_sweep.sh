#!/bin/bash
# The measurement set of a build, in one call on one box (run from the repo root on the GPU box):  scripts/final_sweep.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-r06}
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
bash scripts/pmc_traffic.sh ${tag}_pmc > $O/pmc.log 2>&1
cp gpurun_out/${tag}_pmc_traffic.json profiles/${tag}_traffic.json && cp gpurun_out/${tag}_pmc_traffic.json $O/traffic.json
python bench.py 2>/dev/null | grep '^{' > $O/bench_n1.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/bench.py --no-cpu-baseline --record-files 0 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_L200 -- python $R/bench.py --no-cpu-baseline --record-files 0 --long-seq 200 --attn-dtype fp8 > /dev/null 2>&1
cd $R
python scripts/kernel_summary.py $O/prof/*/*_kernel_trace.csv > $O/kernel_summary.txt
python scripts/kernel_summary.py $O/prof_L200/*/*_kernel_trace.csv > $O/kernel_summary_L200.txt
cp $O/prof/*/*_kernel_stats.csv $O/kernel_stats.csv; cp $O/prof_L200/*/*_kernel_stats.csv $O/kernel_stats_L200.csv
rm -rf $O/prof/*/*_kernel_trace.csv $O/prof_L200/*/*_kernel_trace.csv
python bench.py --no-cpu-baseline --record-files 0 --long-seq 200 --attn-dtype fp8 2>/dev/null | grep '^{' > $O/bench_n1_L200_fp8.json
python bench.py --no-cpu-baseline --record-files 0 --long-seq 200 2>/dev/null | grep '^{' > $O/bench_n1_L200_bf16.json
python bench.py --no-cpu-baseline --record-files 0 --long-seq 200 --batch 8192 2>/dev/null | grep '^{' > $O/bench_n1_L200_bf16_b8192.json
python bench.py --no-cpu-baseline --record-files 0 --long-seq 200 --attn-dtype fp8 --batch 8192 2>/dev/null | grep '^{' > $O/bench_n1_L200_fp8_b8192.json
python bench.py --no-cpu-baseline --record-files 0 --lengths ragged 2>/dev/null | grep '^{' > $O/bench_n1_ragged.json
DMT_PACKED_ROWS=0 python bench.py --no-cpu-baseline --record-files 0 --lengths ragged 2>/dev/null | grep '^{' > $O/bench_n1_ragged_dense_layout.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_ragged -- python $R/bench.py --no-cpu-baseline --record-files 0 --lengths ragged > /dev/null 2>&1
cd $R
python scripts/kernel_summary.py $O/prof_ragged/*/*_kernel_trace.csv > $O/kernel_summary_ragged.txt
rm -rf $O/prof_ragged/*/*_kernel_trace.csv
python bench.py --no-cpu-baseline --record-files 0 --law uniform 2>/dev/null | grep '^{' > $O/bench_n1_uniform.json
python bench.py --no-cpu-baseline --record-files 0 --fresh-batches 4 --age-tables 0 2>/dev/null | grep '^{' > $O/bench_n1_round2_protocol.json
DMT_DETERMINISTIC=1 python bench.py --no-cpu-baseline --record-files 0 2>/dev/null | grep '^{' > $O/bench_n1_deterministic.json
DMT_BENCH_FORCE_DP=1 python bench.py --no-cpu-baseline --record-files 0 2>/dev/null | grep '^{' > $O/bench_forced_dp.json
DMT_BENCH_FORCE_DP=1 python bench.py --no-cpu-baseline --record-files 0 --shard-tables --sku-rows 100000000 2>/dev/null | grep '^{' > $O/bench_forced_dp_sharded_100m.json
for f in $O/bench*.json; do python -c "
import json,sys; d=json.load(open('$f')); print('$f'.split('/')[-1], d['ms_per_step'], d['value'], d['roofline'].get('key'), d['roofline'].get('frac'), d['roofline'].get('avg_launch_us'))"; done
head -8 $O/kernel_summary.txt

#!/bin/bash
# L2 (TCC) request / hit / miss counts of chain2_kernel, ONE counter per rocprofv3 pass (combining TCC counters hung nodes on this pool):
#   scripts/pmc_chain_l2.sh <out-name>     (run on the GPU box from the repo root)
R=${GRAFT_REPO_ROOT:-$(pwd)}
name=${1:-r3chain_l2}
cd /tmp && export TMPDIR=/tmp
for c in TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum FETCH_SIZE WRITE_SIZE; do
  (cd $R && timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/${name}_$c -- python scripts/chain_micro.py > $R/gpurun_out/${name}_$c.log 2>&1) || echo "pass $c failed"
done
cd $R
python - <<PY
import csv, glob, collections
for c in ("TCC_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum", "FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob("gpurun_out/${name}_%s/*/*counter_collection.csv" % c)
    if not fs:
        print(c, "no file"); continue
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(fs[0])):
        if "chain2_kernel" not in r["Kernel_Name"] or r["Counter_Name"] != c: continue
        mode = r["Kernel_Name"].split("Geo<320, 1280, 320>, ")[1][:1]
        agg.setdefault(mode, []).append(float(r["Counter_Value"]))
    for mode, v in agg.items():
        # launches in issue order: 13 forward(train) + 13 forward(infer) for mode 0, 13 backward for mode 1
        if mode == "0":
            print(c, "fwd train avg %.5g   fwd infer avg %.5g" % (sum(v[:13]) / 13, sum(v[13:26]) / max(len(v[13:26]), 1)))
        else:
            print(c, "bwd avg %.5g" % (sum(v) / len(v)))
PY

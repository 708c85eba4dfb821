#!/bin/bash
# SQ / LDS counters of the kernels whose name contains <pattern>:   scripts/pmc_kernel.sh <pattern> <out-name> <command...>
# (run on the GPU box from the repo root; three rocprofv3 --pmc passes of <= 8 SQ counters each, kernel trace only)
R=${GRAFT_REPO_ROOT:-$(pwd)}
pat=$1; name=$2; shift 2
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA"
P2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS"
P3="GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_WAVES"
i=0
for p in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  (cd $R && timeout 300 rocprofv3 --kernel-trace --pmc $p --output-format csv -d $R/gpurun_out/${name}_p$i -- "$@" > $R/gpurun_out/${name}_p$i.log 2>&1) || echo "pass $i failed"
done
cd $R
python - <<PY
import csv, glob, collections
agg = collections.OrderedDict()
for i in (1, 2, 3):
    for f in glob.glob("gpurun_out/${name}_p%d/*/*counter_collection.csv" % i):
        for r in csv.DictReader(open(f)):
            if "$pat" not in r["Kernel_Name"]: continue
            k = (r["Kernel_Name"].split("(")[0][-40:] + " grid " + r["Grid_Size"], r["Counter_Name"])
            agg.setdefault(k, []).append(float(r["Counter_Value"]))
last = None
for (kn, cn), v in agg.items():
    if kn != last: print("##", kn); last = kn
    print("  %-28s n=%-3d avg=%.5g" % (cn, len(v), sum(v) / len(v)))
PY

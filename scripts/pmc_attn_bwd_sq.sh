# SQ counters of the short attention backward (scripts/attn_one.py), one counter per rocprofv3 --pmc pass (round 6: what saturates at 5+ wavefronts per CU?)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
dirs=""
for c in GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC; do
  timeout 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/sqb_$c -- python $R/scripts/attn_one.py 2 > $R/gpurun_out/sqb_$c.log 2>&1 || echo "pass $c failed"
  dirs="$dirs,gpurun_out/sqb_$c"
done
cd $R
python - <<PY
import csv,glob,collections
for d in "$dirs".strip(",").split(","):
    fs=glob.glob(d+"/*/*counter_collection.csv")
    if not fs: print(d,"none"); continue
    agg=collections.OrderedDict()
    for r in csv.DictReader(open(fs[0])):
        n=r["Kernel_Name"]
        if "attn" not in n: continue
        agg.setdefault(("fwd" if "fwd" in n else "bwd", r["Counter_Name"]),[]).append(float(r["Counter_Value"]))
    for k,v in agg.items(): print("%s %-28s %.4g"%(k[0],k[1],sum(v)/len(v)))
PY

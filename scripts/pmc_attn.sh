R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for c in "TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_GATE_EN1_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "TCP_TA_TCP_STATE_READ_sum TA_FLAT_WRITE_WAVEFRONTS_sum" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TD_TD_BUSY_sum"; do
  i=$((i+1))
  timeout 150 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pa$i -- python $R/scripts/attn_one.py 2 > $R/gpurun_out/pa$i.log 2>&1 || echo "pass $i failed"
done
cd $R
python scripts/pmc_summary.py gpurun_out/pa1,gpurun_out/pa2,gpurun_out/pa3,gpurun_out/pa4,gpurun_out/pa5,gpurun_out/pa6 attn

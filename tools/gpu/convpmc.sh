mkdir -p gpurun_out/convpmc
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCC_HIT_sum TCC_MISS_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCC_TAG_STALL_sum TCC_BUSY_avr TCC_LATENCY_FIFO_FULL_sum TCC_SRC_FIFO_FULL_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_LATENCY_sum TA_BUSY_avr" "GRBM_GUI_ACTIVE TCP_GATE_EN1_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum TCC_READ_sum TCC_CYCLE_sum"; do
  i=$((i+1))
  for v in old new; do
    if [ $v = old ]; then export AVEC_NO_CONV_SHIFT=1; else unset AVEC_NO_CONV_SHIFT; fi
    timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/cp_${i}_$v -o p -- python $R/tools/conv_pmc_target.py > /tmp/cp_${i}_$v.log 2>&1
    f=$(find /tmp/cp_${i}_$v -name "*counter_collection.csv" | head -1)
    if [ -n "$f" ]; then cp $f $R/gpurun_out/convpmc/set${i}_$v.csv; else tail -5 /tmp/cp_${i}_$v.log > $R/gpurun_out/convpmc/set${i}_$v.err; fi
  done
done

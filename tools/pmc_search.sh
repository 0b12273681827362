# usage: bash tools/pmc_search.sh  -> gpurun_out/pmc_sq*/ ; summarised by tools/pmc_summary.py
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_FLAT" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" ; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_sq$i -- python3 $R/bench.py --steps 2 --warmup 1 --pairs 4000000 --no-cpu-baseline ${BENCH_ARGS} > $R/gpurun_out/pmc_sq$i.log 2>&1 || tail -3 $R/gpurun_out/pmc_sq$i.log
done
cd $R; python tools/pmc_summary.py

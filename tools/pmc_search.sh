# usage: bash tools/pmc_search.sh  -> gpurun_out/pmc_sq*/ ; summarised by tools/pmc_summary.py
# every pass runs under its own timeout: a counter set the profiler cannot schedule must not eat the budget
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_sq*
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_LDS SQ_INSTS_BRANCH" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" ; do
  i=$((i+1))
  timeout 90 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_sq$i -- python3 $R/bench.py --steps 2 --warmup 1 --pairs 4000000 --no-cpu-baseline ${BENCH_ARGS} > $R/gpurun_out/pmc_sq$i.log 2>&1 || echo "pass $i failed/timeout"
done
cd $R; python tools/pmc_summary.py

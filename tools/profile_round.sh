# usage (on the GPU box): bash tools/profile_round.sh <tag> [extra bench.py args]   -> gpurun_out/<tag>_*
# kernel-trace stats of the seeding-stage leg of bench.py + separate PMC passes (each under its own timeout)
TAG=${1:-r02}; shift; X="$@"; R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
python3 $R/bench.py --leg seeding --seed-steps 1 --pairs 1000000 $X > /dev/null 2>&1   # build + cache the index outside the profiler
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_trace -- python3 $R/bench.py --leg seeding --seed-steps 5 $X > $R/gpurun_out/${TAG}_trace.log 2>&1 || echo "trace pass failed"
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH"; do
  n=$(echo $set | cut -d" " -f1)
  timeout 400 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/${TAG}_pmc_$n -- python3 $R/bench.py --leg seeding --seed-steps 2 $X > $R/gpurun_out/${TAG}_pmc_$n.log 2>&1 || echo "pmc pass $n failed"
done
cd $R; echo done

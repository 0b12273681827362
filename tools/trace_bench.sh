# usage (GPU box): bash tools/trace_bench.sh <tag> [pairs]  -> gpurun_out/<tag>_kernel_stats.csv : rocprofv3 kernel trace of bench.py (2 mapping runs)
TAG=${1:-trace}; PAIRS=${2:-10000000}; R=$GRAFT_REPO_ROOT
A="--pairs $PAIRS --steps 1 --warmup 1 --no-cpu-baseline --no-parity --no-seeding-leg --no-other-configs"
python3 $R/bench.py --pairs 1000000 --leg seeding --seed-steps 1 > /dev/null 2>&1
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_trace -- python3 $R/bench.py $A > $R/gpurun_out/${TAG}_trace.log 2>&1 || echo "trace pass failed"
f=$(find $R/gpurun_out/${TAG}_trace -name "*kernel_stats.csv" | head -1); cp "$f" $R/gpurun_out/${TAG}_kernel_stats.csv
rm -rf $R/gpurun_out/${TAG}_trace
grep -E "aln_|sam_|fq_|search_kernel|nw_|chain|sort_small|locate" $R/gpurun_out/${TAG}_kernel_stats.csv | cut -d, -f1-4 | cut -c1-150

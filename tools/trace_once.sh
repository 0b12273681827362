# usage (GPU box): bash tools/trace_once.sh <tag> [pairs] -- kernel trace of one kart-amd run on the hg38-sized index (env passes through)
TAG=${1:-t}; PAIRS=${2:-2000000}; R=$GRAFT_REPO_ROOT
WD=$(python3 -c "import sys; sys.path.insert(0,'$R'); import bench; print(bench.pick_workdir(60<<30))")
[ -f $WD/l1.fq ] || E2E_NO_REF=1 python3 $R/tools/e2e_large.py 3100000000 $PAIRS > /dev/null 2>&1
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_trace -- $R/kart_amd/bin/kart-amd -silent -i $WD/synth_v2_3100000000 -f $WD/l1.fq -f2 $WD/l2.fq -o $WD/prof.sam -t 32 > $R/gpurun_out/${TAG}_trace.log 2>&1
f=$(find $R/gpurun_out/${TAG}_trace -name "*kernel_stats.csv" | head -1); cp "$f" $R/gpurun_out/${TAG}_kernel_stats.csv
grep -E "aln_|search_kernel|nw_" $R/gpurun_out/${TAG}_kernel_stats.csv | cut -d, -f1-4 | cut -c1-120

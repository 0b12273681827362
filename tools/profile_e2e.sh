# usage (GPU box): bash tools/profile_e2e.sh <tag> [pairs]   -> gpurun_out/<tag>_e2e_kernel_stats.csv
# rocprofv3 kernel trace of one kart-amd FASTQ -> SAM run on the hg38-sized index of bench.py (the binary itself after "--")
TAG=${1:-r02}; PAIRS=${2:-2000000}; R=$GRAFT_REPO_ROOT
python3 $R/tools/e2e_large.py 3100000000 $PAIRS > $R/gpurun_out/${TAG}_e2e_plain.json 2>/dev/null      # builds the index + the FASTQ files, plain run
WD=$(python3 -c "import sys; sys.path.insert(0,'$R'); import bench; print(bench.pick_workdir(60<<30))")
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_e2e_trace -- $R/kart_amd/bin/kart-amd -silent -i $WD/synth_v2_3100000000 -f $WD/l1.fq -f2 $WD/l2.fq -o $WD/prof.sam -t 32 > $R/gpurun_out/${TAG}_e2e_trace.log 2>&1
f=$(find $R/gpurun_out/${TAG}_e2e_trace -name "*kernel_stats.csv" | head -1)
cp "$f" $R/gpurun_out/${TAG}_e2e_kernel_stats.csv
head -30 $R/gpurun_out/${TAG}_e2e_kernel_stats.csv | cut -c1-200

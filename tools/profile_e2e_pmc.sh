# usage (GPU box): bash tools/profile_e2e_pmc.sh <tag> [pairs]   -> gpurun_out/<tag>_e2e_*  (kernel trace + separate PMC passes of one kart-amd run)
TAG=${1:-r02}; PAIRS=${2:-2000000}; R=$GRAFT_REPO_ROOT
E2E_NO_REF=1 python3 $R/tools/e2e_large.py 3100000000 $PAIRS > $R/gpurun_out/${TAG}_e2e_plain.json 2>/dev/null      # builds the index + the FASTQ files, plain run
WD=$(python3 -c "import sys; sys.path.insert(0,'$R'); import bench; print(bench.pick_workdir(60<<30))")
cd /tmp; export TMPDIR=/tmp
CMD="$R/kart_amd/bin/kart-amd -silent -i $WD/synth_v2_3100000000 -f $WD/l1.fq -f2 $WD/l2.fq -o $WD/prof.sam -t 32"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_e2e_trace -- $CMD > $R/gpurun_out/${TAG}_e2e_trace.log 2>&1
f=$(find $R/gpurun_out/${TAG}_e2e_trace -name "*kernel_stats.csv" | head -1); cp "$f" $R/gpurun_out/${TAG}_e2e_kernel_stats.csv
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  n=$(echo $set | cut -d" " -f1)
  timeout 600 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/${TAG}_e2e_pmc_$n -- $CMD > $R/gpurun_out/${TAG}_e2e_pmc_$n.log 2>&1 || echo "pmc pass $n failed"
done
cd $R
python3 - <<PY
import collections, csv, glob, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/${TAG}_e2e_pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("void ", "").split("(")[0]
        if "kg::" in name: agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: {"dispatches": len(v), "sum": sum(v)} for c, v in cs.items()} for k, cs in agg.items()}
json.dump(out, open("gpurun_out/${TAG}_e2e_pmc_summary.json", "w"), indent=1, sort_keys=True)
print("kernels with counters:", len(out))
PY
head -24 gpurun_out/${TAG}_e2e_kernel_stats.csv | cut -c1-160

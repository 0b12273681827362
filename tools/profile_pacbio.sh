# usage (GPU box): bash tools/profile_pacbio.sh [reads] -- per-kernel time of the configs[3] (-pacbio) run: rocprofv3 --kernel-trace --stats over kart-amd itself
cd $GRAFT_REPO_ROOT
N=${1:-200000}
RUN_CONFIGS_NO_REF=1 RUN_CONFIGS_KEEP_INPUTS=1 python3 tools/run_configs.py $N 0 > gpurun_out/pb_keep.json 2> /dev/null
CMD=$(python3 -c "import json; print(' '.join(json.load(open('gpurun_out/pb_keep.json'))['configs[3] -pacbio']['command']))")
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_pb
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
KART_AMD_VERBOSE=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o pb -- $CMD > $OUT/run.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/pb_kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel time in total: %.1f ms" % (tot / 1e6))
for r in rows[:16]:
    print("%-60s calls %6s total %9.2f ms avg %9.1f us  %5.1f %%" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
grep -E "mapping seconds|stage seconds|fragment" $OUT/run.log | cut -c1-250

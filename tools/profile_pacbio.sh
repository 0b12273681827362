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
# [reads] pmc : two counter passes over the same command, summed per kernel
if [ "$2" = "pmc" ]; then
  cd /tmp
  i=0
  for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVES"; do
    i=$((i+1))
    rocprofv3 --pmc $set --output-format csv -d $OUT/pmc$i -o pb -- $CMD > $OUT/pmc$i.log 2>&1 || echo "pmc pass $i failed"
  done
  cd $GRAFT_REPO_ROOT
  python3 - <<PY
import collections, csv, glob, json
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for f in glob.glob("$OUT/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "").split("(")[0]
        if "kg::" in k: agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
out = {k: dict(v) for k, v in agg.items()}
json.dump(out, open("gpurun_out/pb_pmc_summary.json", "w"), indent=1, sort_keys=True)
for k in ("kg::frag_partition_kernel", "kg::search_kernel<unsigned long, false, 0>", "kg::nw_big_kernel<false>", "kg::sort_lds_kernel<2048, 4>", "kg::frag_stitch_kernel"):
    v = out.get(k)
    if not v: continue
    wc = v.get("SQ_WAVE_CYCLES", 1)
    print(k, "| wave cycles %.3g" % wc, "| issue %.0f %%" % (100 * v.get("SQ_ACTIVE_INST_ANY", 0) / wc), "| wait any %.0f %%" % (100 * v.get("SQ_WAIT_ANY", 0) / wc), "| wait inst %.0f %%" % (100 * v.get("SQ_WAIT_INST_ANY", 0) / wc),
          "| VALU %.3g SALU %.3g LDS %.3g VMEM rd %.3g wr %.3g" % (v.get("SQ_INSTS_VALU", 0), v.get("SQ_INSTS_SALU", 0), v.get("SQ_INSTS_LDS", 0), v.get("SQ_INSTS_VMEM_RD", 0), v.get("SQ_INSTS_VMEM_WR", 0)),
          "| LDS bank conflict cycles %.3g, LDS active %.3g, VALU active %.3g, busy %.3g, waves %.3g" % (v.get("SQ_LDS_BANK_CONFLICT", 0), v.get("SQ_ACTIVE_INST_LDS", 0), v.get("SQ_ACTIVE_INST_VALU", 0), v.get("SQ_BUSY_CYCLES", 0), v.get("SQ_WAVES", 0)))
PY
fi

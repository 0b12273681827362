# usage (GPU box): bash tools/profile_nw.sh <tag>   -> gpurun_out/<tag>_nw_*
# the NW kernels alone (tools/bench_nw.py, device-resident batches per size class): kernel trace + one PMC pass
TAG=${1:-r02}; R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
python3 $R/tools/bench_nw.py > $R/gpurun_out/${TAG}_nw_bench.json 2>/dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_nw_trace -- python3 $R/tools/bench_nw.py > $R/gpurun_out/${TAG}_nw_trace.log 2>&1 || echo "trace pass failed"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $R/gpurun_out/${TAG}_nw_pmc -- python3 $R/tools/bench_nw.py > $R/gpurun_out/${TAG}_nw_pmc.log 2>&1 || echo "pmc pass failed"
cd $R
python3 - <<PY
import collections, csv, glob, json
tag = "$TAG"
out = {"bench": json.load(open("gpurun_out/%s_nw_bench.json" % tag))}
st = sorted(glob.glob("gpurun_out/%s_nw_trace/**/*kernel_stats.csv" % tag, recursive=True))
if st:
    out["kernel_stats"] = [{k: r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs")} for r in csv.DictReader(open(st[-1])) if "nw_" in r["Name"]]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/%s_nw_pmc/**/*counter_collection.csv" % tag, recursive=True)):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].replace("void ", "").split("(")[0]
        if "nw_" in n:
            agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
out["pmc_mean_per_launch"] = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in agg.items()}
for k, cs in out["pmc_mean_per_launch"].items():
    if cs.get("SQ_WAVE_CYCLES"):
        cs["issue_fraction"] = cs.get("SQ_ACTIVE_INST_ANY", 0) / cs["SQ_WAVE_CYCLES"]
        cs["wait_fraction"] = cs.get("SQ_WAIT_INST_ANY", 0) / cs["SQ_WAVE_CYCLES"]
json.dump(out, open("gpurun_out/%s_nw_profile.json" % tag, "w"), indent=1)
print(json.dumps(out["bench"]))
for k, cs in out["pmc_mean_per_launch"].items():
    print(k, {c: round(v, 3) if v < 10 else int(v) for c, v in cs.items()})
PY

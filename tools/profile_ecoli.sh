# usage (GPU box): bash tools/profile_ecoli.sh <tag> [pairs]   -> gpurun_out/<tag>_ecoli_*
# BASELINE.json configs[1] (E. coli-sized index, 150 bp pairs): the bench on that workload, a kernel trace, and the L2 hit / miss + read-request
# counters of its kernels (the 9 MB rank structure lives in the L2 / Infinity Cache -- report L2-hit counters there instead of an HBM
# fraction).  The summary's `_l2` entry is what bench.py's other_configs["configs[1]"] quotes.
TAG=${1:-r05}; PAIRS=${2:-10000000}; R=$GRAFT_REPO_ROOT
A="--genome-len 4639675 --pairs $PAIRS --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-seeding-leg --no-other-configs"
python3 $R/bench.py $A > $R/gpurun_out/${TAG}_ecoli_bench_plain.json 2>/dev/null
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_ecoli_trace -- python3 $R/bench.py $A > $R/gpurun_out/${TAG}_ecoli_trace.log 2>&1 || echo "trace pass failed"
f=$(find $R/gpurun_out/${TAG}_ecoli_trace -name "*kernel_stats.csv" | head -1); cp "$f" $R/gpurun_out/${TAG}_ecoli_kernel_stats.csv
for set in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum"; do
  n=$(echo $set | cut -d" " -f1)
  timeout 900 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/${TAG}_ecoli_pmc_$n -- python3 $R/bench.py $A > $R/gpurun_out/${TAG}_ecoli_pmc_$n.log 2>&1 || echo "pmc pass $n failed"
done
cd $R
python3 - <<PY
import collections, csv, glob, json
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("gpurun_out/${TAG}_ecoli_pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("void ", "").split("(")[0]
        if "kg::" in name: agg[name.replace("kg::", "").split("<")[0]][r["Counter_Name"]] += float(r["Counter_Value"])
out = {k: dict(v) for k, v in agg.items()}
l2 = {}
for k, v in out.items():
    h, m = v.get("TCC_HIT_sum", 0), v.get("TCC_MISS_sum", 0)
    if h + m > 0:
        l2[k] = {"TCC_HIT": h, "TCC_MISS": m, "hit_rate": h / (h + m), "hbm_read_bytes": v.get("TCC_EA0_RDREQ_128B_sum", 0) * 128 + v.get("TCC_EA0_RDREQ_64B_sum", 0) * 64}
tot_h = sum(x["TCC_HIT"] for x in l2.values()); tot_m = sum(x["TCC_MISS"] for x in l2.values())
out["_l2"] = {"pairs_per_step": $PAIRS, "steps_profiled": 3, "search_kernel": l2.get("search_kernel"), "all_kernels_hit_rate": tot_h / (tot_h + tot_m) if tot_h + tot_m else None,
              "per_kernel_hit_rate": {k: round(x["hit_rate"], 4) for k, x in sorted(l2.items(), key=lambda kv: -kv[1]["TCC_HIT"] - kv[1]["TCC_MISS"])[:14]},
              "note": "rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum (one pass) and TCC_EA0_RDREQ_* (another) over bench.py --genome-len 4639675, summed over the run's launches"}
json.dump(out, open("gpurun_out/${TAG}_ecoli_pmc_summary.json", "w"), indent=1, sort_keys=True)
print(json.dumps(out["_l2"], indent=1)[:1500])
PY
head -16 gpurun_out/${TAG}_ecoli_kernel_stats.csv | cut -c1-160
rm -rf gpurun_out/${TAG}_ecoli_trace gpurun_out/${TAG}_ecoli_pmc_TCC_HIT_sum gpurun_out/${TAG}_ecoli_pmc_TCC_EA0_RDREQ_128B_sum

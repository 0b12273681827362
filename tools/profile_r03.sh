# usage (GPU box): bash tools/profile_r03.sh <tag> [pairs]   -> gpurun_out/<tag>_bench_*
# rocprofv3 kernel trace + separate PMC passes of bench.py itself (the FASTQ -> SAM steps of the timed region; python3 after "--")
TAG=${1:-r03}; PAIRS=${2:-10000000}; R=$GRAFT_REPO_ROOT
A="--pairs $PAIRS --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-seeding-leg --no-other-configs"
python3 $R/bench.py $A > $R/gpurun_out/${TAG}_bench_plain.json 2>/dev/null       # builds + caches the index; the unprofiled line
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_bench_trace -- python3 $R/bench.py $A > $R/gpurun_out/${TAG}_bench_trace.log 2>&1 || echo "trace pass failed"
f=$(find $R/gpurun_out/${TAG}_bench_trace -name "*kernel_stats.csv" | head -1); cp "$f" $R/gpurun_out/${TAG}_bench_kernel_stats.csv
for set in "TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  n=$(echo $set | cut -d" " -f1)
  timeout 900 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/${TAG}_bench_pmc_$n -- python3 $R/bench.py $A > $R/gpurun_out/${TAG}_bench_pmc_$n.log 2>&1 || echo "pmc pass $n failed"
done
cd $R
python3 - <<PY
import collections, csv, glob, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/${TAG}_bench_pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("void ", "").split("(")[0]
        if "kg::" in name: agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: {"dispatches": len(v), "sum": sum(v)} for c, v in cs.items()} for k, cs in agg.items()}
# search_kernel's fabric traffic over all its launches of the run (warm-up + timed steps alike: same launches every step)
for k, cs in list(out.items()):
    if "search_kernel" in k and "TCC_EA0_RDREQ_128B_sum" in cs:
        n = cs["TCC_EA0_RDREQ_128B_sum"]["dispatches"]
        rd = cs["TCC_EA0_RDREQ_128B_sum"]["sum"] * 128 + cs.get("TCC_EA0_RDREQ_64B_sum", {"sum": 0})["sum"] * 64
        wr = cs.get("WRITE_SIZE", {"sum": 0, "dispatches": n})
        wr_b = wr["sum"] * 1024 * (n / max(1, wr["dispatches"]))
        out["_search_traffic"] = {"tag": "timed", "kernel": k.replace("kg::", ""), "genome_len": 3100000000, "launches": n, "pairs_per_step": $PAIRS,
                                  "read_bytes_corrected": rd, "WRITE_SIZE_bytes": wr_b, "traffic_bytes_per_launch": (rd + wr_b) / n,
                                  "note": "gfx950: read traffic = RDREQ_128B x 128 + RDREQ_64B x 64 (FETCH_SIZE tallies 128-B requests at 64 B, MI355X_MICROARCH.md HBM section); "
                                          "WRITE_SIZE in KiB; separate passes; per launch = the sum over the run's launches / their number"}
json.dump(out, open("gpurun_out/${TAG}_bench_pmc_summary.json", "w"), indent=1, sort_keys=True)
print("kernels with counters:", len(out))
PY
head -30 gpurun_out/${TAG}_bench_kernel_stats.csv | cut -c1-170

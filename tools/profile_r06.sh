# usage (GPU box): bash tools/profile_r06.sh <tag> [pairs]   -> gpurun_out/<tag>_bench_*
# rocprofv3 kernel trace + separate PMC passes of bench.py itself (the FASTQ -> SAM steps of the timed region; python3 straight after "--").
# Round 6: the aln_trivial slot; the gpu_pipeline leg is left out of the profiled command.
# Round 5: the summary also holds `_kernel_traffic` -- counter bytes per step of every kernel bench.py prices (its `kernels` entries).
# Round 4: the default workload size (100 M reads per step), the L2 hit / miss pass is back, and the summary records the configuration
# (pairs per step, seeding group) so that bench.py only quotes `traffic` for the very configuration that was profiled.
TAG=${1:-r06}; PAIRS=${2:-50000000}; R=$GRAFT_REPO_ROOT
A="--pairs $PAIRS --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-seeding-leg --no-other-configs --no-gpu-pipeline"
python3 $R/bench.py $A > $R/gpurun_out/${TAG}_bench_plain.json 2>/dev/null       # builds + caches the index; the unprofiled line
cd /tmp; export TMPDIR=/tmp
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_bench_trace -- python3 $R/bench.py $A > $R/gpurun_out/${TAG}_bench_trace.log 2>&1 || echo "trace pass failed"
f=$(find $R/gpurun_out/${TAG}_bench_trace -name "*kernel_stats.csv" | head -1); cp "$f" $R/gpurun_out/${TAG}_bench_kernel_stats.csv
for set in "TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  n=$(echo $set | cut -d" " -f1)
  timeout 1200 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/${TAG}_bench_pmc_$n -- python3 $R/bench.py $A > $R/gpurun_out/${TAG}_bench_pmc_$n.log 2>&1 || echo "pmc pass $n failed"
done
cd $R
python3 - <<PY
import collections, csv, glob, json, os
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/${TAG}_bench_pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("void ", "").split("(")[0]
        if "kg::" in name: agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: {"dispatches": len(v), "sum": sum(v)} for c, v in cs.items()} for k, cs in agg.items()}
plain = [json.loads(l) for l in open("gpurun_out/${TAG}_bench_plain.json") if l.startswith("{")]
cfg = plain[-1]["config"] if plain else {}
for k, cs in list(out.items()):
    if "search_kernel" in k and "TCC_EA0_RDREQ_128B_sum" in cs:
        n = cs["TCC_EA0_RDREQ_128B_sum"]["dispatches"]
        rd = cs["TCC_EA0_RDREQ_128B_sum"]["sum"] * 128 + cs.get("TCC_EA0_RDREQ_64B_sum", {"sum": 0})["sum"] * 64
        wr = cs.get("WRITE_SIZE", {"sum": 0, "dispatches": n})
        wr_b = wr["sum"] * 1024 * (n / max(1, wr["dispatches"]))
        hit, miss = cs.get("TCC_HIT_sum", {"sum": 0})["sum"], cs.get("TCC_MISS_sum", {"sum": 0})["sum"]
        out["_search_traffic"] = {"tag": "timed", "kernel": k.replace("kg::", ""), "genome_len": 3100000000, "launches": n, "pairs_per_step": $PAIRS,
                                  "seed_group": cfg.get("seed_group"), "stream_lanes": cfg.get("stream_lanes"), "stream_reads": cfg.get("stream_reads"), "sa_mode": cfg.get("sa_mode", "full"),
                                  "read_bytes_corrected": rd, "WRITE_SIZE_bytes": wr_b, "traffic_bytes_per_launch": (rd + wr_b) / n,
                                  "l2_hit_rate": hit / (hit + miss) if hit + miss else None, "TCC_HIT": hit, "TCC_MISS": miss,
                                  "note": "gfx950: read traffic = RDREQ_128B x 128 + RDREQ_64B x 64 (FETCH_SIZE tallies 128-B requests at 64 B, MI355X_MICROARCH.md HBM section); "
                                          "WRITE_SIZE in KiB; separate passes; per launch = the sum over the run's launches / their number (ramp-up launches included, as in roofline.achieved)"}
# per-kernel counter traffic per step, for the "kernels" entries of the bench line (the passes ran steps + warmup = 3 steps)
slots = {"chain": ["chain_kernel"], "aln_trivial": ["aln_trivial_kernel"], "aln_pair": ["aln_pair_kernel"], "aln_rescue": ["aln_rescue_kernel", "aln_post_rescue_kernel"], "aln_plan_fast": ["aln_plan_fast_kernel"],
         "aln_plan": ["aln_plan_kernel"], "aln_partition": ["aln_partition_kernel"], "nw": ["nw_small8_kernel", "nw_small32_kernel", "nw_big_kernel", "nw_classify_kernel"],
         "aln_finish": ["aln_finish_kernel", "aln_finish_wave_kernel", "aln_finish_group_kernel"], "aln_final": ["aln_final_kernel"], "sam_size": ["sam_size_kernel"], "sam_format": ["sam_format_kernel"],
         "fq_parse": ["fq_count_kernel", "fq_index_kernel", "fq_record_kernel", "fq_plan_kernel", "fq_reset_kernel"], "fq_materialise": ["fq_materialise_kernel"]}
STEPS = 3
kt = {}
for slot, names in slots.items():
    rd = wr = 0.0
    hit = miss = 0.0
    for k, cs in out.items():
        base = k.replace("kg::", "").split("<")[0]
        if base not in names or "TCC_EA0_RDREQ_128B_sum" not in cs: continue
        rd += cs["TCC_EA0_RDREQ_128B_sum"]["sum"] * 128 + cs.get("TCC_EA0_RDREQ_64B_sum", {"sum": 0})["sum"] * 64
        wr += cs.get("WRITE_SIZE", {"sum": 0})["sum"] * 1024
        hit += cs.get("TCC_HIT_sum", {"sum": 0})["sum"]; miss += cs.get("TCC_MISS_sum", {"sum": 0})["sum"]
    if rd + wr > 0: kt[slot] = {"bytes": (rd + wr) / STEPS, "read": rd / STEPS, "written": wr / STEPS, "l2_hit_rate": hit / (hit + miss) if hit + miss else None}
out["_kernel_traffic"] = {"pairs_per_step": $PAIRS, "genome_len": 3100000000, "seed_group": cfg.get("seed_group"), "stream_lanes": cfg.get("stream_lanes"), "stream_reads": cfg.get("stream_reads"), "sa_mode": cfg.get("sa_mode", "full"),
                          "steps_profiled": STEPS, "bytes_per_step": {k: v["bytes"] for k, v in kt.items()}, "detail": kt,
                          "note": "per kernel: (RDREQ_128B x 128 + RDREQ_64B x 64 + WRITE_SIZE KiB x 1024) summed over the run's launches / the 3 steps the passes ran"}
json.dump(out, open("gpurun_out/${TAG}_bench_pmc_summary.json", "w"), indent=1, sort_keys=True)
print("kernels with counters:", len(out), out.get("_search_traffic"))
PY
head -30 gpurun_out/${TAG}_bench_kernel_stats.csv | cut -c1-170
# the raw traces and counter files are far beyond what travels back (64 MiB): keep the condensed files only
rm -rf gpurun_out/${TAG}_bench_trace gpurun_out/${TAG}_bench_pmc_TCC_EA0_RDREQ_128B_sum gpurun_out/${TAG}_bench_pmc_WRITE_SIZE gpurun_out/${TAG}_bench_pmc_TCC_HIT_sum gpurun_out/${TAG}_bench_pmc_SQ_WAVE_CYCLES

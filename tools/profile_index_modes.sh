# usage (GPU box): bash tools/profile_index_modes.sh [modes...] -- kernel trace + one PMC pass of the seeding leg (20 M reads per launch) per index mode
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
python3 bench.py --pairs 10000000 --leg seeding --seed-steps 1 > /dev/null 2>&1      # builds + caches the index files
cd /tmp; export TMPDIR=/tmp
for sa in ${@:-compact dense4}; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/idx_$sa -o t -- python3 $R/bench.py --pairs 10000000 --leg seeding --seed-steps 3 --sa $sa > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $R/gpurun_out/idx_${sa}_pmc -o p -- python3 $R/bench.py --pairs 10000000 --leg seeding --seed-steps 3 --sa $sa > /dev/null 2>&1
done
cd $R
python3 - "$@" <<'PY'
import collections, csv, glob, json, sys
modes = sys.argv[1:] or ["compact", "dense4"]
out = {}
for sa in modes:
    st = glob.glob("gpurun_out/idx_%s/**/t_kernel_stats.csv" % sa, recursive=True)
    rows = [r for r in csv.DictReader(open(st[0])) if any(k in r["Name"] for k in ("search_kernel", "locate_", "sort_", "pack_reads", "finish_offsets"))] if st else []
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob("gpurun_out/idx_%s_pmc/**/p_counter_collection.csv" % sa, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("void ", "").split("(")[0]
            if "search_kernel" in k or "locate_" in k: agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    out[sa] = {"kernel_stats": [{k: r[k] for k in ("Name", "Calls", "AverageNs")} for r in rows], "pmc_sum": {k: dict(v) for k, v in agg.items()}}
    for k, v in agg.items():
        wc = v.get("SQ_WAVE_CYCLES", 1)
        print(sa, k, "| issue %.0f %% | wait any %.0f %% | VMEM rd %.3g" % (100 * v.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * v.get("SQ_WAIT_ANY", 0) / wc, v.get("SQ_INSTS_VMEM_RD", 0)))
    for r in rows: print(sa, r["Name"][:60], r["Calls"], "avg %.2f ms" % (float(r["AverageNs"]) / 1e6))
json.dump(out, open("gpurun_out/r03y_index_modes_profile.json", "w"), indent=1)
PY
rm -rf gpurun_out/idx_*

R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
python3 $R/bench.py --genome-len 1000000000 --pairs 4000000 --steps 1 --warmup 0 --no-cpu-baseline --no-e2e > /dev/null 2>&1   # builds + caches the index
for set in "TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $set | cut -d" " -f1)
  timeout 200 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/g1_pmc_$n -- python3 $R/bench.py --genome-len 1000000000 --pairs 4000000 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e > $R/gpurun_out/g1_pmc_$n.log 2>&1 || echo "pmc pass $n failed"
done
cd $R; python - <<'PY'
import csv, glob, collections, json
agg = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/g1_pmc_*/**/*_counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "search_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {c: sum(v)/len(v) for c, v in agg.items()}
print(json.dumps(out))
json.dump(out, open("gpurun_out/g1_pmc_summary.json", "w"))
PY
timeout 900 python tools/e2e_compare.py 500000 1000000000 | tee gpurun_out/e2e_1g.json

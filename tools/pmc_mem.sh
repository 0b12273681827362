R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "TCC_EA0_RDREQ[A-Z0-9_a-z]*|TCC_EA0_RD_[A-Za-z0-9_]*|TCC_REQ[A-Za-z0-9_]*|TCC_READ[A-Za-z0-9_]*|TCC_BUBBLE[A-Za-z0-9_]*|TCC_TAG_STALL[A-Za-z0-9_]*|MALL[A-Za-z0-9_]*" | sort -u | tr '\n' ' ' > $R/gpurun_out/tcc_counters.txt
i=0
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_READ_sum" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_mem$i -- python3 $R/bench.py --steps 2 --warmup 1 --pairs 4000000 --no-cpu-baseline --genome-len ${GLEN:-400000000} > $R/gpurun_out/pmc_mem$i.log 2>&1 || echo "pass $i failed"
done
cd $R; cat gpurun_out/tcc_counters.txt; echo; python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/pmc_mem*/**/*_counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "search_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(agg.items()): print(f"{c:30s} n={len(v)} mean={sum(v)/len(v):.4g}")
PY

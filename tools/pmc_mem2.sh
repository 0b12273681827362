R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
timeout 120 rocprofv3 --pmc TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum --output-format csv -d $R/gpurun_out/pmc_memx -- python3 $R/bench.py --steps 2 --warmup 1 --pairs 4000000 --no-cpu-baseline --genome-len ${GLEN:-400000000} > $R/gpurun_out/pmc_memx.log 2>&1 || echo "pass failed"
timeout 120 rocprofv3 --pmc TCC_EA0_RDREQ_DRAM_sum TCC_READ_SECTORS_sum --output-format csv -d $R/gpurun_out/pmc_memy -- python3 $R/bench.py --steps 2 --warmup 1 --pairs 4000000 --no-cpu-baseline --genome-len ${GLEN:-400000000} > $R/gpurun_out/pmc_memy.log 2>&1 || echo "pass failed"
cd $R; python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list)
for f in sorted(glob.glob("gpurun_out/pmc_mem[xy]/**/*_counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "search_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(agg.items()): print(f"{c:30s} n={len(v)} mean={sum(v)/len(v):.4g}")
PY

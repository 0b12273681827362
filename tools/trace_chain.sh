# usage (GPU box): bash tools/trace_chain.sh -- kernel trace of bench.py (20 M reads per step) under lane / group variations: per-kernel average of the stage kernels
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
A="--pairs 10000000 --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-seeding-leg --no-other-configs"
python3 $R/bench.py $A > /dev/null 2>&1
for spec in "0:4:" "0:8:" "4:8:KG_GROUP_NO_TURNS=1" "4:8:"; do
  IFS=: read g l e <<< "$spec"
  export KART_AMD_SEED_GROUP=$g KART_AMD_STREAM_LANES=$l
  [ -n "$e" ] && export $e
  rm -rf /tmp/tc; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tc -- python3 $R/bench.py $A > /tmp/tc.log 2>&1
  [ -n "$e" ] && unset ${e%%=*}
  f=$(find /tmp/tc -name "*kernel_stats.csv" | head -1)
  echo "== group $g lanes $l $e"
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].replace("void ", "").split("(")[0]
    if n in ("kg::chain_kernel", "kg::compact_cands_kernel", "kg::aln_pair_kernel", "kg::aln_plan_kernel", "kg::sam_format_kernel", "kg::aln_rescue_kernel", "kg::fq_materialise_kernel") or "search_kernel" in n:
        print("   %-34s calls %4s  total %8.1f ms  avg %7.3f ms  max %7.3f ms" % (n[:34], r["Calls"], int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6, int(r["MaxNs"]) / 1e6))
PY
done

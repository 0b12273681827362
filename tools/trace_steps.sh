# usage (GPU box): bash tools/trace_steps.sh [pairs] -- rocprofv3 kernel trace of bench.py's timed steps (2 steps): per-kernel totals, per step
R=$GRAFT_REPO_ROOT; PAIRS=${1:-10000000}; cd /tmp; export TMPDIR=/tmp
A="--pairs $PAIRS --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-seeding-leg --no-other-configs"
python3 $R/bench.py $A > /dev/null 2>&1
rm -rf /tmp/tb; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tb -- python3 $R/bench.py $A > /tmp/tb.log 2>&1
f=$(find /tmp/tb -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
tot = 0
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].replace("void ", "").split("(")[0]
    if n.startswith("kg::") and not any(k in n for k in ("expand_sa", "qtab", "planes_k", "build_", "sample_sa")):
        ms = int(r["TotalDurationNs"]) / 3e6          # 3 mapping runs (1 warm-up + 2 steps)
        tot += ms
        if ms > 1.0:
            print("   %-36s calls %5s  %7.1f ms per step  avg %7.3f ms" % (n[:36], r["Calls"], ms, float(r["AverageNs"]) / 1e6))
print("   sum of kg:: kernels: %.0f ms per step" % tot)
PY

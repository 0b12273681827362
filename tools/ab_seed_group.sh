#!/bin/bash
# tools/ab_seed_group.sh OUTDIR [PAIRS] -- bench.py's timed steps under seeding-group variations, alternating on one box:
#   "group lanes" pairs, e.g. "0 4" = four independent lanes (round 3), "4 8" = two groups of four lanes (default)
set -u
OUT=$1; PAIRS=${2:-50000000}
mkdir -p $OUT
B="python3 bench.py --gpus 1 --steps ${STEPS:-4} --warmup 1 --pairs $PAIRS --no-other-configs --no-parity --no-cpu-baseline --no-seeding-leg"
for rep in 1 2; do
	for spec in ${SPECS:-"0:4" "4:8" "4:12" "6:12" "8:16"}; do
		g=${spec%%:*}; l=${spec##*:}
		KART_AMD_SEED_GROUP=$g KART_AMD_STREAM_LANES=$l timeout 900 $B > $OUT/ab_g${g}_l${l}_$rep.log 2> $OUT/ab_g${g}_l${l}_$rep.err
		python3 - <<PY
import json
try:
    d=[json.loads(l) for l in open("$OUT/ab_g${g}_l${l}_$rep.log") if l.startswith("{")][-1]
    r=d["roofline"]
    print("group $g lanes $l run $rep: %.2f M mapped reads/s" % (d["value"]/1e6), [round(x,3) for x in d["rank0_step_seconds"]], "frac %.4f, %d launches, avg %.3f ms, %.0f reads per launch" % (r["frac"], r["launches"], r["avg_launch_ms"], d["config"]["reads_per_step"]*d["steps"]/r["launches"]), "peak shmem", d["config"]["peak_shmem_GB"])
except Exception as e:
    print("group $g lanes $l run $rep: failed", e)
PY
	done
done

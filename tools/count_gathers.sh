# usage (GPU box): bash tools/count_gathers.sh [extra bench args]  -- per-read tallies of the search kernel's gathers by kind
for t in 1 2 3 4; do
  KG_DEBUG_COUNT=$t python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-e2e "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); w=d['work_per_read']
print({1:'table lookups',2:'LF steps executed',3:'LF steps with two 128-byte lines',4:'text-compare rounds'}[$t], round(w['inv'],2), '| searches', round(w['searches'],2), 'sa gathers', round(w['sa'],2), 'lf1+lf2 (reference steps)', round(w['lf1']+w['lf2'],1))"
done

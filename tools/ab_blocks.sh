# usage (GPU box): bash tools/ab_blocks.sh "8 4 8 4" [extra bench args] -- alternating runs of KG_SEARCH_BLOCKS_PER_CU
LIST=$1; shift
for b in $LIST; do
  KG_SEARCH_BLOCKS_PER_CU=$b python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-e2e "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('blocks/CU', $b, round(d['value']/1e6,1), 'Mreads/s search ms', round(d['kernels_ms']['search'],2))"
done

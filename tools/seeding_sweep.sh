# usage (GPU box): bash tools/seeding_sweep.sh "<env assignments, ';'-separated>" [bench args]   e.g.  "KG_SEARCH_BLOCKS_PER_CU=4;KG_SEARCH_BLOCKS_PER_CU=8" --pairs 4000000
# runs the seeding-stage leg of bench.py (GPU seeding step on resident reads) once per setting, alternating, and prints its kernel times
LIST=$1; shift
IFS=';' read -ra SETS <<< "$LIST"
for s in "${SETS[@]}"; do
  env $s python bench.py --leg seeding --seed-steps 5 "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())['seeding_stage']; r=d['roofline']
print('$s', round(d['value']/1e6,1), 'Mreads/s', {k:round(v,2) for k,v in d['kernels_ms'].items()}, 'useful GB/s', round(r['achieved']), 'frac', round(r['frac'],3), {k:round(v,2) for k,v in d['fetched_per_read'].items()})"
done

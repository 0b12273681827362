# usage (GPU box): bash tools/ab_occupancy.sh -- search kernel time of bench.py's seeding leg for several blocks-per-CU settings
run() { timeout 600 python bench.py --leg seeding --seed-steps 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['seeding_stage']['kernels_ms'], d['seeding_stage'].get('oracle_sample'))"; }
for b in 4 5 6 4 5; do KG_SEARCH_BLOCKS_PER_CU=$b run blocks$b; done

for b in 1 2 4 6 8; do echo "blocks_per_cu=$b"; KG_SEARCH_BLOCKS_PER_CU=$b python bench.py --steps 3 --warmup 1 --pairs 4000000 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys;d=json.loads(sys.stdin.read());print(round(d['value']/1e6,1),'Mreads/s',d['kernels_ms'])"; done

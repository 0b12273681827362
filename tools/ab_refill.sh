# usage (GPU box): bash tools/ab_refill.sh -- search kernel time for several refill thresholds (rebuilds the library in the box's copy)
run() { timeout 600 python bench.py --leg seeding --seed-steps 5 $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['seeding_stage']['kernels_ms']['search'])"; }
for r in 44 32 52 60 44; do sed -i "s/constexpr int kRefill = [0-9]*;/constexpr int kRefill = $r;/" kart_amd/csrc/kernels/search.inc; make -C kart_amd/csrc -j8 > /dev/null 2>&1
  run "refill$r"
done
for b in 5 8; do KG_SEARCH_BLOCKS_PER_CU=$b run "refill44 blocks$b"; done

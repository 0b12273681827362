# 1 Gbp with both builders: files must be identical; then timing of the bucketed one
set -e
cd $GRAFT_REPO_ROOT
python bench.py --genome-len 1000000000 --pairs 2000000 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e 2>&1 | tail -1 | cut -c1-400
python bench.py --genome-len 1000000000 --pairs 2000000 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --bucketed 2>&1 | grep -v "^\[" | tail -1 | cut -c1-400
D=/tmp/kart_bench_$(id -u)
for e in bwt sa pac ann amb; do cmp $D/synth_1000000000.$e $D/synth_1000000000_b.$e && echo "$e identical"; done

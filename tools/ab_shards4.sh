# usage (GPU box): bash tools/ab_shards4.sh -- 1 / 2 / 4 shard processes on ONE device with -parts (500 Mbp genome so that four index replicas fit):
# a later shard writing its text while it maps (round 4) against holding it back until it has settled (KART_AMD_NO_EAGER_PARTS=1, round 3)
cd $GRAFT_REPO_ROOT
A="--genome-len 500000000 --pairs 10000000 --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-seeding-leg --no-other-configs"
python3 bench.py --genome-len 500000000 --pairs 1000000 --leg seeding --seed-steps 1 > /dev/null 2>&1
run() { label=$1; n=$2; shift 2; env "$@" KART_BENCH_SHARE_DEVICE=1 KART_AMD_STREAM_LANES=2 KART_AMD_SEED_GROUP=2 python3 bench.py --gpus $n $A --parts > gpurun_out/shards4.log 2>&1
  echo "== $label: $(grep -o '"value": [0-9.]*' gpurun_out/shards4.log | tail -1) $(grep -o '"rank0_step_seconds": [^]]*]' gpurun_out/shards4.log | tail -1) $(grep -o '"chunks_remapped_per_step": [0-9.]*' gpurun_out/shards4.log | tail -1)"; }
for rep in 1 2; do
run "1 process" 1 X=1
run "2 processes, -parts, text written while mapping" 2 X=1
run "2 processes, -parts, text held back (round 3)" 2 KART_AMD_NO_EAGER_PARTS=1
run "4 processes, -parts, text written while mapping" 4 X=1
run "4 processes, -parts, text held back (round 3)" 4 KART_AMD_NO_EAGER_PARTS=1
done

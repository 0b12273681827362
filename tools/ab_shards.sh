# usage (GPU box): bash tools/ab_shards.sh -- the sharding overhead on ONE device: the same 20 M reads through 1, 2 and 4 processes that share device 0
# (500 Mbp synthetic genome, so that four index replicas and their lanes fit one MI355X; same total thread budget; bench.py's own multi-rank path, gloo)
cd $GRAFT_REPO_ROOT
A="--genome-len 500000000 --pairs 10000000 --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-seeding-leg --no-other-configs"
python bench.py --genome-len 500000000 --pairs 1000000 --leg seeding --seed-steps 1 > /dev/null 2>&1     # builds + caches the index
for n in 1 2 4 1 4; do
  KART_BENCH_SHARE_DEVICE=1 KART_AMD_VERBOSE=1 KART_AMD_STREAM_LANES=2 python bench.py --gpus $n $A > gpurun_out/shards_$n.log 2>&1
  echo "== $n process(es): $(grep -o '"value": [0-9.]*' gpurun_out/shards_$n.log | head -1) $(grep -o '"rank0_step_seconds": [^]]*]' gpurun_out/shards_$n.log) $(grep -o '"chunks_remapped_per_step": [0-9.]*' gpurun_out/shards_$n.log) $(grep -o '"sam_bytes_per_step": [0-9]*' gpurun_out/shards_$n.log)"
  grep -E "^shard [0-9]" gpurun_out/shards_$n.log | tail -$n
done

# usage (GPU box): bash tools/ab_shards5.sh -- why do 2 shard processes on ONE device reach 60-70 % of one process while 4 reach 91-100 %?
# Hypothesis: batches in flight.  Every process keeps KART_AMD_STREAM_LANES batches in flight; the device runs one process's kernels at a time.
cd $GRAFT_REPO_ROOT
A="--genome-len 500000000 --pairs 10000000 --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-seeding-leg --no-other-configs"
python3 bench.py --genome-len 500000000 --pairs 1000000 --leg seeding --seed-steps 1 > /dev/null 2>&1
run() { label=$1; n=$2; shift 2; env "$@" KART_BENCH_SHARE_DEVICE=1 python3 bench.py --gpus $n $A --parts > gpurun_out/shards5.log 2>&1
  echo "== $label: $(grep -o '"value": [0-9.]*' gpurun_out/shards5.log | tail -1) $(grep -o '"rank0_step_seconds": [^]]*]' gpurun_out/shards5.log | tail -1)"; }
run "1 process, 2 lanes" 1 KART_AMD_STREAM_LANES=2 KART_AMD_SEED_GROUP=0
run "1 process, 4 lanes" 1 KART_AMD_STREAM_LANES=4 KART_AMD_SEED_GROUP=0
run "1 process, 8 lanes in groups of 4 (default)" 1 X=1
run "2 processes x 2 lanes" 2 KART_AMD_STREAM_LANES=2 KART_AMD_SEED_GROUP=0
run "2 processes x 4 lanes" 2 KART_AMD_STREAM_LANES=4 KART_AMD_SEED_GROUP=0
run "2 processes x 8 lanes in groups of 4 (default)" 2 X=1
run "4 processes x 2 lanes" 4 KART_AMD_STREAM_LANES=2 KART_AMD_SEED_GROUP=0
run "4 processes x 4 lanes" 4 KART_AMD_STREAM_LANES=4 KART_AMD_SEED_GROUP=0
run "4 processes x 8 lanes in groups of 4 (default)" 4 X=1

# usage (GPU box): bash tools/ab_shards2.sh -- 4 shard processes on one device: where does the overhead come from?
cd $GRAFT_REPO_ROOT
A="--genome-len 500000000 --pairs 10000000 --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-seeding-leg --no-other-configs"
python bench.py --genome-len 500000000 --pairs 1000000 --leg seeding --seed-steps 1 > /dev/null 2>&1
run() { label=$1; n=$2; extra=$3; shift 3; env "$@" KART_BENCH_SHARE_DEVICE=1 KART_AMD_STREAM_LANES=2 python bench.py --gpus $n $A $extra > gpurun_out/shards2.log 2>&1
  echo "== $label: $(grep -o '"value": [0-9.]*' gpurun_out/shards2.log | head -1) $(grep -o '"rank0_step_seconds": [^]]*]' gpurun_out/shards2.log)"; }
run "1 process" 1 "" X=1
run "4 processes, one file, taking turns (default)" 4 "--one-file" X=1
run "4 processes, one file, no turns" 4 "--one-file" KART_AMD_NO_FILE_TURNS=1
run "4 processes, one file per process (-parts)" 4 "--parts" X=1
run "2 processes, one file" 2 "--one-file" X=1
run "2 processes, -parts" 2 "--parts" X=1
run "1 process again" 1 "" X=1

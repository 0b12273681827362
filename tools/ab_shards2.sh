# usage (GPU box): bash tools/ab_shards2.sh -- 4 shard processes on one device: where does the overhead come from?
cd $GRAFT_REPO_ROOT
A="--genome-len 500000000 --pairs 10000000 --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-seeding-leg --no-other-configs"
python bench.py --genome-len 500000000 --pairs 1000000 --leg seeding --seed-steps 1 > /dev/null 2>&1
run() { label=$1; n=$2; shift 2; env "$@" KART_BENCH_SHARE_DEVICE=1 KART_AMD_STREAM_LANES=2 python bench.py --gpus $n $A > gpurun_out/shards2.log 2>&1
  echo "== $label: $(grep -o '"value": [0-9.]*' gpurun_out/shards2.log | head -1) $(grep -o '"rank0_step_seconds": [^]]*]' gpurun_out/shards2.log)"; }
run "1 process" 1 X=1
run "4 processes, I/O threads of each on its own L3" 4 X=1
run "4 processes, I/O threads of all on ONE L3 (0-7,128-135)" 4 KART_AMD_IO_CPUS=0-7,128-135
run "4 processes, not pinned" 4 KART_AMD_IO_CPUS=none
run "4 processes, one writer thread each" 4 KART_AMD_WRITER_THREADS=1
run "2 processes, all on one L3" 2 KART_AMD_IO_CPUS=0-7,128-135

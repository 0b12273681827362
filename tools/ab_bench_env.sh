# usage (GPU box): bash tools/ab_bench_env.sh -- bench.py (20 M reads, hg38-sized, session reused over the steps) under environment variations
cd $GRAFT_REPO_ROOT
B="python bench.py --pairs ${PAIRS:-10000000} --steps ${STEPS:-5} --warmup 1 --no-cpu-baseline --no-parity --no-seeding-leg --no-other-configs"
show() { echo "== $2: $(grep -o '"value": [0-9.]*' $1 | head -1) $(grep -o '"frac": [0-9.]*' $1 | head -1) $(grep -o '"search_kernel_ms_per_step": [0-9.]*' $1) $(grep -o '"device_ms_per_step": {[^w]*' $1) $(grep -o '"rank0_step_seconds": [^]]*]' $1)"; grep -E "cpu seconds" $1 | tail -3 | cut -c1-120; grep -E "^stream:|^i/o threads" $1 | tail -2 | cut -c1-330; }
run() { label=$1; shift; env "$@" KART_AMD_VERBOSE=1 $B > gpurun_out/ab_$$.log 2>&1; show gpurun_out/ab_$$.log "$label"; }
if [ -n "$1" ]; then
  while [ -n "$1" ]; do run "$1" ${1//,/ }; shift; done      # (several assignments: A=1,B=2)
  exit 0
fi
run "default (pinned, prealloc)" X=1
run "no pin" KART_AMD_IO_CPUS=none
run "pinned, no prealloc" KART_AMD_NO_PREALLOC=1
run "pinned, writers 6" KART_AMD_WRITER_THREADS=6
run "pinned, writers 2" KART_AMD_WRITER_THREADS=2
run "pinned, lanes 4" KART_AMD_STREAM_LANES=4
run "pinned, batch 2M" KART_AMD_STREAM_READS=2000000
run "default again" X=1
rm -f gpurun_out/ab_$$.log

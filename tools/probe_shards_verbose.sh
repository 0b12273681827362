# usage (GPU box): bash tools/probe_shards_verbose.sh -- kart-amd -gpu 0,0[,0,0] -parts on ONE device with KART_AMD_VERBOSE: where does a shard process spend its time?
cd $GRAFT_REPO_ROOT
A="--genome-len 500000000 --pairs 10000000 --steps 1 --warmup 0 --no-cpu-baseline --no-parity --no-seeding-leg --no-other-configs"
KART_BENCH_KEEP_FASTQ=1 python3 bench.py $A > /dev/null 2>&1
D=$(ls -d /dev/shm/kart_bench_*); P=$(ls $D/synth_v2_500000000*.bwt | head -1); P=${P%.bwt}
for g in 0 0,0 0,0,0,0; do
  for rep in 1 2; do
    echo "== -gpu $g (run $rep)"
    ( time KART_AMD_VERBOSE=1 kart_amd/bin/kart-amd -silent -i $P -f $D/bench_1.fq -f2 $D/bench_2.fq -o $D/probe.sam -gpu $g -parts -t 16 ) 2>&1 | grep -E "^shard|^real|stage seconds|cpu seconds|device stream:|All the" | cut -c1-330
    rm -f $D/probe.sam*
  done
done
rm -f $D/bench_1.fq $D/bench_2.fq

# usage (GPU box): bash tools/ab_stream.sh -- kart-amd FASTQ -> SAM (20 M reads, hg38-sized index) through the device stream under
# variations of the output side (writer threads, page pre-allocation, /dev/null) and of the stream (lanes, batch size)
cd $GRAFT_REPO_ROOT
E2E_NO_REF=1 timeout 900 python tools/e2e_large.py 3100000000 10000000 > /dev/null 2>&1
WD=$(python3 -c "import sys; sys.path.insert(0,'.'); import bench; print(bench.pick_workdir(60<<30))")
ls -la $WD | head; df -h /dev/shm | tail -1; free -g | head -2
run() { # label, output, env...
  label=$1; out=$2; shift 2
  [ "$out" != "keep" ] && rm -f $WD/o.sam
  [ "$out" = "keep" ] && out=$WD/o.sam
  env "$@" KART_AMD_VERBOSE=1 kart_amd/bin/kart-amd -silent -i $WD/synth_v2_3100000000 -f $WD/l1.fq -f2 $WD/l2.fq -o $out -t 16 | grep -E "mapping seconds|cpu seconds|^stream:" | cut -c1-260 | tr '\n' ' '; echo " <- $label"
}
for i in 1 2 3; do
  run "default" $WD/o.sam X=1
  run "null output" /dev/null X=1
  run "no prealloc" $WD/o.sam KART_AMD_NO_PREALLOC=1
  run "writers 2" $WD/o.sam KART_AMD_WRITER_THREADS=2
  run "writers 8" $WD/o.sam KART_AMD_WRITER_THREADS=8
  run "reuse output file" keep X=1
  run "lanes 4" $WD/o.sam KART_AMD_STREAM_LANES=4
  run "batch 2M" $WD/o.sam KART_AMD_STREAM_READS=2000000
done

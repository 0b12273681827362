# usage (GPU box): bash tools/ab_pacbio_batch.sh [reads] -- kart-amd -pacbio on the hg38-sized index for several batch sizes (chunks of 10 reads)
N=${1:-100000}; R=$GRAFT_REPO_ROOT
bash $R/tools/trace_pacbio.sh tq $N > /dev/null 2>&1      # builds index + reads (and one traced run)
WD=$(python3 -c "import sys; sys.path.insert(0,'$R'); import bench; print(bench.pick_workdir(60<<30))")
for c in 1024 1024; do
  KART_AMD_PACBIO_CHUNKS=$c KART_AMD_VERBOSE=1 $R/kart_amd/bin/kart-amd -silent -i $WD/synth_v2_3100000000 -f $WD/pb.fq -pacbio -o $WD/pb.sam -t 32 | grep -E "mapping seconds|worker thread-seconds" | sed "s/^/chunks $c: /" | cut -c1-260
done

# the long-read report on the device (kg_longread_batch), on the GPU box: the pacbio golden with KART_AMD_CHECK_ALIGN, then the
# focused pytest cases; everything into gpurun_out/$1_*
tag=${1:-r05b}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out /tmp/lr && cd /tmp/lr
B=$GRAFT_REPO_ROOT/kart_amd/bin/kart-amd
G=$GRAFT_REPO_ROOT/tests/golden
zcat $G/sam/pacbio.fq.gz > pb.fq
zcat $G/sam/pacbio.sam.gz > want.sam
for env in "KART_AMD_VERBOSE=1" "KART_AMD_VERBOSE=1 KART_AMD_CHECK_ALIGN=1" "KART_AMD_VERBOSE=1 KART_AMD_HOST_LONG=1"; do
  echo "== $env"
  env $env timeout 300 $B -i $G/idx/small -f pb.fq -pacbio -o got.sam -t 4 2>&1 | grep -E "device report|CHECK_ALIGN|Error|error|long-read" | cut -c1-900 | head -40
  cmp got.sam want.sam && echo "golden pacbio: identical"
done
diff <(cut -f1-9 got.sam) <(cut -f1-9 want.sam) | head -20
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_sam_gpu.py -x -q -k "pacbio or long" 2>&1 | tail -15

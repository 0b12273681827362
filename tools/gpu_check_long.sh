# the long-read report on the device (kg_longread_batch), on the GPU box: the pacbio golden with KART_AMD_CHECK_ALIGN, then the
# focused pytest cases; everything into gpurun_out/$1_*
tag=${1:-r05b}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out /tmp/lr && cd /tmp/lr
B=$GRAFT_REPO_ROOT/kart_amd/bin/kart-amd
G=$GRAFT_REPO_ROOT/tests/golden
zcat $G/sam/pacbio.fq.gz > pb.fq
zcat $G/sam/pacbio.sam.gz > want.sam
KART_AMD_VERBOSE=1 KG_LONG_DEBUG=${DBG_READ:-0} timeout 300 $B -i $G/idx/small -f pb.fq -pacbio -o got.sam -t 4 > run.log 2>&1
grep -E "device report: [0-9]|long-read" run.log | cut -c1-600
grep -A200 "KG_LONG_DEBUG" run.log | head -${DBG_LINES:-150} | cut -c1-1500
cmp got.sam want.sam && echo "golden pacbio: identical"
python3 - <<PY
g = [l.split("\t") for l in open("got.sam") if not l.startswith("@")]
w = [l.split("\t") for l in open("want.sam") if not l.startswith("@")]
bad = [i for i, (x, y) in enumerate(zip(g, w)) if x != y]
print(len(g), len(w), "records;", len(bad), "differ:", bad[:20])
for i in bad[:3]:
    x, y = g[i], w[i]
    print(i, x[0], [k for k in range(len(x)) if k >= len(y) or x[k] != y[k]])
    print("  got ", x[1:5], x[5][:700])
    print("  want", y[1:5], y[5][:700])
PY
if [ -n "$RUN_TESTS" ]; then cd $GRAFT_REPO_ROOT; timeout 900 python -m pytest tests/test_sam_gpu.py -x -q -k "pacbio or long" 2>&1 | tail -15; fi

# the one unexpected difference of tools/fuzz_on_gpu_box.sh in round 5 (profiles/r05a_fuzz_on_box.log): plain FASTQ with 1500-character
# headers through the device stream -- what does the product say?
cd $GRAFT_REPO_ROOT; mkdir -p /tmp/lh && cd /tmp/lh
python3 - <<PY
import sys
sys.path.insert(0, "$GRAFT_REPO_ROOT")
from kart_amd import synth
from kart_amd.index_build import read_fasta
genome = {n: s for n, _, s in read_fasta("$GRAFT_REPO_ROOT/tests/golden/small.fa")}
names, reads = synth.simulate_long_reads(genome, 40, seed=3, read_len=1200, err=0.02)
synth.write_fastq("l12.fq", names, reads)
synth.write_fastq("lh.fq", [n + " " + "x" * 1500 for n in names], reads)
PY
for f in l12.fq lh.fq; do
  echo "== $f"
  KART_AMD_VERBOSE=1 $GRAFT_REPO_ROOT/kart_amd/bin/kart-amd -i $GRAFT_REPO_ROOT/tests/golden/idx/small -f $f -o lh.sam -t 4 > lh.log 2>&1; echo "status $?"
  grep -v "^stage\|^worker\|^cpu" lh.log | tail -8 | cut -c1-400; wc -l lh.sam
done

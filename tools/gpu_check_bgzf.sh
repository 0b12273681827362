#!/bin/bash
# tools/gpu_check_bgzf.sh -- bgzip-ped input on the box, CLI only: the golden pairs as BGZF against the golden SAM, then 270 k pairs (the
# golden pairs thirty times over) as BGZF with the members inflated side by side and through one gzread() stream (KART_AMD_NO_BGZF=1).
B=kart_amd/bin/kart-amd
S=tests/golden/idx/small
O=${TMPDIR:-/tmp}
mkdir -p gpurun_out
{
python3 - <<PY
import gzip, sys
sys.path.insert(0, "tests")
from bgzf_util import bgzf
for m in ("1", "2"):
    raw = gzip.open("tests/golden/sam/pe_%s.fq.gz" % m).read()
    open("$O/pe_%s.b.fq.gz" % m, "wb").write(bgzf(raw))
    open("$O/big_%s.b.fq.gz" % m, "wb").write(bgzf(raw * 30, level=1))
PY
timeout 40 $B -silent -i $S -f $O/pe_1.b.fq.gz -f2 $O/pe_2.b.fq.gz -t 16 -o $O/b.sam; zcat tests/golden/sam/pe.sam.gz | cmp - $O/b.sam && echo OK golden pairs as BGZF
for rep in 1 2; do
echo "== members side by side"; KART_AMD_VERBOSE=1 timeout 60 $B -i $S -f $O/big_1.b.fq.gz -f2 $O/big_2.b.fq.gz -t 16 -o $O/n.sam | grep -E "mapping seconds|read_batch total|cpu seconds" | cut -c1-200
echo "== one gzread() stream (KART_AMD_NO_BGZF=1)"; KART_AMD_NO_BGZF=1 KART_AMD_VERBOSE=1 timeout 60 $B -i $S -f $O/big_1.b.fq.gz -f2 $O/big_2.b.fq.gz -t 16 -o $O/s.sam | grep -E "mapping seconds|read_batch total|cpu seconds" | cut -c1-200
done
cmp $O/n.sam $O/s.sam && echo outputs identical
bash tools/gpu_check_hostpath.sh | grep "^OK"
} > gpurun_out/bgzf_check.log 2>&1
cat gpurun_out/bgzf_check.log

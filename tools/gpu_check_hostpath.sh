#!/bin/bash
# tools/gpu_check_hostpath.sh -- the product CLI (no Python, no torch: a few seconds per run) on inputs whose kart -t 1 output was
# made beforehand in the build container (tests/_build/gpucheck/, see tools/README.md): the pacbio golden, reads with literal '-' / N,
# 3000 x 7 kb reads with pass 2 from the op strings and from the gapped strings.  Prints OK<n> per identical output.
B=kart_amd/bin/kart-amd
D=tests/_build/gpucheck
S=tests/golden/idx/small
O=${TMPDIR:-/tmp}
mkdir -p gpurun_out
{
timeout 40 $B -silent -i $S -f tests/golden/sam/pacbio.fq.gz -pacbio -t 16 -o $O/o1.sam; zcat tests/golden/sam/pacbio.sam.gz | cmp - $O/o1.sam && echo OK1 pacbio golden
timeout 40 $B -silent -i $S -f $D/d_long.fq -pacbio -t 16 -o $O/o2.sam; cmp $O/o2.sam $D/d_long.ref.sam && echo OK2 long reads with dashes
timeout 40 $B -silent -i $S -f $D/d_1.fq -f2 $D/d_2.fq -t 16 -o $O/o3.sam; cmp $O/o3.sam $D/d_short.ref.sam && echo OK3 short pairs with dashes
KART_AMD_VERBOSE=1 timeout 60 $B -i $D/g -f $D/long.fq -pacbio -t 16 -o $O/o4.sam | grep -E "thread-seconds|fragment pairs|mapping seconds|cpu seconds"; cmp $O/o4.sam $D/long.ref.sam && echo OK4 3000 x 7 kb, pass 2 from the op strings
KART_AMD_FINISH_STRINGS=1 KART_AMD_VERBOSE=1 timeout 60 $B -i $D/g -f $D/long.fq -pacbio -t 16 -o $O/o5.sam | grep -E "thread-seconds|fragment pairs|mapping seconds|cpu seconds"; cmp $O/o5.sam $D/long.ref.sam && echo OK5 3000 x 7 kb, pass 2 from the gapped strings
} > gpurun_out/hostpath_check.log 2>&1
cat gpurun_out/hostpath_check.log

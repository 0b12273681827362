#!/bin/bash
# tools/gpu_check_pass2_scale.sh -- pass 2 of the -pacbio report from the op strings against the gapped strings at 60 000 x 7 kb reads on
# the box (the 3000 reads of tests/_build/gpucheck twenty times over; 5 Mbp genome, so seeding is lighter than on hg38: the host
# sections are what this shows).  CLI only.
B=kart_amd/bin/kart-amd
D=tests/_build/gpucheck
O=${TMPDIR:-/tmp}
mkdir -p gpurun_out
{
for i in $(seq 20); do cat $D/long.fq; done > $O/long60k.fq
for rep in 1 2; do
echo "== op strings"; KART_AMD_VERBOSE=1 timeout 60 $B -i $D/g -f $O/long60k.fq -pacbio -t 16 -o $O/n.sam | grep -E "thread-seconds|mapping seconds|cpu seconds|stage seconds"
echo "== gapped strings (KART_AMD_FINISH_STRINGS=1)"; KART_AMD_FINISH_STRINGS=1 KART_AMD_VERBOSE=1 timeout 60 $B -i $D/g -f $O/long60k.fq -pacbio -t 16 -o $O/s.sam | grep -E "thread-seconds|mapping seconds|cpu seconds|stage seconds"
done
cmp $O/n.sam $O/s.sam && echo outputs identical
head -c $(stat -c %s $D/long.ref.sam) $O/n.sam | cmp - $D/long.ref.sam && echo "first 3000 reads identical to kart -t 1"
} > gpurun_out/pass2_scale.log 2>&1
cat gpurun_out/pass2_scale.log

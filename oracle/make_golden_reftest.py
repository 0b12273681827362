#!/usr/bin/env python3
"""configs[0] as a parser fixture: the 2000 read pairs of the reference's own test (test/r1.fq, test/r2.fq -- data files, run_test.sh:25-27)
mapped by the unmodified reference binary (oracle/_ref/kart -t 1) against the small golden index.  The E. coli FASTA the reference's
test indexes is not in the tree, so most of these reads stay unmapped here: what the fixture pins is the record parsing (headers of the
form "@0:Pos=4267671<TAB>/1": the name ends at the tab) and the unmapped / mapped SAM records of real wgsim-style input.
    python oracle/make_golden_reftest.py     -> tests/golden/sam/ref_test_{r1,r2}.fq.gz, ref_test.sam.gz"""
import gzip, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "sam")
KART = os.path.join(ROOT, "oracle", "_ref", "kart")
PREFIX = os.path.join(ROOT, "tests", "golden", "idx", "small")
src = "/root/reference/test"
for n in ("r1", "r2"):
    with open(os.path.join(src, n + ".fq"), "rb") as fi, gzip.open(os.path.join(OUT, "ref_test_%s.fq.gz" % n), "wb", compresslevel=9) as fo:
        fo.write(fi.read())
sams = []
for perturb in (85, 170):
    out = "/tmp/ref_test_%d.sam" % perturb
    subprocess.run([KART, "-silent", "-t", "1", "-i", PREFIX, "-f", os.path.join(src, "r1.fq"), "-f2", os.path.join(src, "r2.fq"), "-o", out], check=True,
                   env=dict(os.environ, MALLOC_PERTURB_=str(perturb)), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    sams.append(open(out, "rb").read())
assert sams[0] == sams[1], "heap-dependent output"
with gzip.open(os.path.join(OUT, "ref_test.sam.gz"), "wb", compresslevel=9) as fo:
    fo.write(sams[0])
lines = [l for l in sams[0].split(b"\n") if l and not l.startswith(b"@")]
print("ref_test: %d records, %d mapped" % (len(lines), sum(1 for l in lines if l.split(b"\t")[2] != b"*")))

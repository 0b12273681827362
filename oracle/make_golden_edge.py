#!/usr/bin/env python3
"""Edge-case golden SAMs from the unmodified reference (TEST INFRASTRUCTURE; needs oracle/_ref): ragged read lengths
5..300, N-rich / lowercase / IUPAC / all-N reads, homopolymers, unmappable reads, cross-contig and far-apart pairs,
names with spaces, two single-end libraries.  python oracle/make_golden_edge.py"""
import os, sys, subprocess, gzip
sys.path.insert(0, ".")
import numpy as np
from kart_amd import synth
from kart_amd.index_build import read_fasta
G = "tests/golden"; OUT = G + "/sam"
genome = {n: s for n, _, s in read_fasta(G + "/small.fa")}
rng = np.random.default_rng(4242)
chrA, chrB = genome["chrA"], genome["chrB"]
def frag(c, p, L): return c[p:p+L].copy()
recs1, recs2 = [], []
def add(name, a, b): recs1.append((name, a)); recs2.append((name, b))
pos = 5000
for L in (5, 12, 13, 14, 20, 31, 75, 151, 250, 300):
    for rep in range(6):
        p = int(rng.integers(4000, 50000)); ins = max(L, 320)
        a = frag(chrA, p, L); b = synth.revcomp(frag(chrA, p + ins - L, L))
        e = rng.random(L) < 0.02; a[e] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, int(e.sum()))]
        add(f"len{L}_{rep} extra words/1", a, b)
# N-rich, lowercase, IUPAC, all-N, homopolymer
for rep in range(8):
    p = int(rng.integers(4000, 50000)); a = frag(chrA, p, 150); b = synth.revcomp(frag(chrA, p + 250, 150))
    if rep % 4 == 0: a[rng.integers(0, 150, 12)] = ord("N")
    if rep % 4 == 1: a = np.frombuffer(a.tobytes().lower(), np.uint8).copy()
    if rep % 4 == 2: a[[10, 40, 90]] = np.frombuffer(b"RYK", np.uint8)
    if rep % 4 == 3: b[:] = ord("N")
    add(f"odd{rep}/1", a, b)
add("polyA", np.full(150, ord("A"), np.uint8), np.full(150, ord("T"), np.uint8))
add("random_unmappable", np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 150)], np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 150)])
# chimeric pair across contigs, and a pair far apart
add("cross_contig", frag(chrA, 20000, 150), synth.revcomp(frag(chrB, 9000, 150)))
add("far_apart", frag(chrA, 10000, 150), synth.revcomp(frag(chrA, 40000, 150)))
def write(path, recs, mate):
    with open(path, "wb") as fh:
        for nm, r in recs:
            fh.write(b"@" + nm.encode() + b"\n" + r.tobytes() + b"\n+\n" + b"I" * len(r) + b"\n")
write("/tmp/edge_1.fq", recs1, 1); write("/tmp/edge_2.fq", recs2, 2)
KART = "oracle/_ref/kart"; IDX = G + "/idx/small"
def run(args, out, perturb):
    env = dict(os.environ, MALLOC_PERTURB_=str(perturb))
    r = subprocess.run([KART, "-silent", "-t", "1", "-i", IDX] + args + ["-o", out], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return r.returncode, (open(out, "rb").read() if r.returncode == 0 else b"")
cases = {"edge_pe": ["-f", "/tmp/edge_1.fq", "-f2", "/tmp/edge_2.fq"], "edge_se": ["-f", "/tmp/edge_1.fq"], "edge_se_m": ["-f", "/tmp/edge_2.fq", "-m"],
         "edge_multi_lib": ["-f", "/tmp/edge_1.fq", "/tmp/edge_2.fq"]}
for name, args in cases.items():
    rc1, a = run(args, "/tmp/ea.sam", 85); rc2, b = run(args, "/tmp/eb.sam", 170)
    print(name, "rc", rc1, rc2, "lines", a.count(b"\n"), "deterministic", a == b)
    if rc1 == 0 and a == b:
        gzip.open(f"{OUT}/{name}.sam.gz", "wb", 9).write(a)
for f in ("edge_1.fq", "edge_2.fq"):
    gzip.open(f"{OUT}/{f}.gz", "wb", 9).write(open("/tmp/" + f, "rb").read())

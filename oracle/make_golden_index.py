#!/usr/bin/env python3
"""Golden index files from the unmodified reference's own builder (TEST INFRASTRUCTURE; needs oracle/_ref/bwt_index):
a small FASTA with everything the writer has to reproduce byte for byte -- ambiguity runs (N, n, IUPAC, runs of
different characters back to back, at contig start and end), lower case, header comments, a contig whose length is
not a multiple of 4, wrapped and unwrapped lines.  python oracle/make_golden_index.py"""
import os, subprocess, sys
import numpy as np
G = "tests/golden"
rng = np.random.default_rng(77)
def rnd(n): return np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n)].copy()
c1 = rnd(3001); c1[:37] = ord("N"); c1[500:520] = ord("n"); c1[520:533] = ord("R"); c1[1200] = ord("Y"); c1[-9:] = ord("N")
c1[700:900] = np.frombuffer(c1[700:900].tobytes().lower(), np.uint8)
c2 = rnd(2048); c2[1000:1003] = np.frombuffer(b"KMS", np.uint8)
c3 = rnd(777)
c4 = rnd(1500); c4[100:400] = c1[1500:1800]          # a repeat between contigs
with open(G + "/amb.fa", "wb") as fh:
    for name, seq, width in ((b"ctgN first contig with a comment", c1, 60), (b"ctgIUPAC", c2, 70), (b"odd_len\tTAB comment", c3, 10 ** 9), (b"ctgRep", c4, 50)):
        fh.write(b">" + name + b"\n")
        for a in range(0, len(seq), width): fh.write(seq[a:a + width].tobytes() + b"\n")
os.makedirs(G + "/idx_amb", exist_ok=True)
subprocess.run(["oracle/_ref/bwt_index", G + "/amb.fa", G + "/idx_amb/amb"], check=True, stdout=subprocess.DEVNULL)
print(sorted(os.listdir(G + "/idx_amb")))

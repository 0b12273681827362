#!/usr/bin/env python3
"""Golden SAM fixtures from the UNMODIFIED reference mapper (oracle/_ref/kart -t 1).

TEST INFRASTRUCTURE; runs only where /root/reference exists.  Inputs are seeded synthetic reads over
tests/golden/small.fa (index tests/golden/idx/small, built by the reference's bwt_index); every case
is run twice under different MALLOC_PERTURB_ fills and must be byte-identical (otherwise the
reference output depends on uninitialised heap, SURVEY.md App. B-12, and the differing column is
reported).  Output: tests/golden/sam/<case>.sam.gz + the FASTQ/FASTA inputs (gz).

    python oracle/make_golden_sam.py
"""
import gzip
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kart_amd import synth  # noqa: E402
from kart_amd.index_build import read_fasta  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
OUT = os.path.join(GOLD, "sam")
KART = os.path.join(ROOT, "oracle", "_ref", "kart")
PREFIX = os.path.join(GOLD, "idx", "small")


def run_kart(args, out_sam, perturb):
    env = dict(os.environ, MALLOC_PERTURB_=str(perturb))
    subprocess.run([KART, "-silent", "-t", "1", "-i", PREFIX] + args + ["-o", out_sam], check=True, env=env,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return open(out_sam, "rb").read()


def case(name, args, mask_flag=False):
    a = run_kart(args, "/tmp/kart_gold_a.sam", 85)
    b = run_kart(args, "/tmp/kart_gold_b.sam", 170)
    if a != b:
        la, lb = a.split(b"\n"), b.split(b"\n")
        assert len(la) == len(lb), name
        diff_cols = set()
        for x, y in zip(la, lb):
            if x != y:
                fx, fy = x.split(b"\t"), y.split(b"\t")
                diff_cols |= {i for i, (p, q) in enumerate(zip(fx, fy)) if p != q}
        assert mask_flag and diff_cols == {1}, (name, diff_cols)
        print(f"  {name}: FLAG column is heap-dependent on some records (App. B-12); stored from the first run")
    with gzip.open(os.path.join(OUT, name + ".sam.gz"), "wb", compresslevel=9) as fh:
        fh.write(a)
    print(f"  {name}: {a.count(10)} lines")


def gz_copy(src, dst):
    with open(src, "rb") as fi, gzip.open(dst, "wb", compresslevel=9) as fo:
        fo.write(fi.read())


def main():
    assert os.path.exists(KART), "build oracle/_ref first (make -C oracle ref)"
    os.makedirs(OUT, exist_ok=True)
    genome = {n: s for n, _, s in read_fasta(os.path.join(GOLD, "small.fa"))}
    # paired-end: 4500 pairs = 9000 reads = 2.25 chunks of 4000 -> EstDistance switches from 1500 to the estimate
    names, r1, r2 = synth.simulate_pairs(genome, 4500, seed=101, err=0.01, n_frac=0.0005)
    f1, f2 = os.path.join(OUT, "pe_1.fq"), os.path.join(OUT, "pe_2.fq")
    synth.write_fastq(f1, names, r1, mate=1)
    synth.write_fastq(f2, names, r2, mate=2)
    case("pe", ["-f", f1, "-f2", f2])
    case("pe_m", ["-f", f1, "-f2", f2, "-m"], mask_flag=True)
    # interleaved (-p) input, 1200 pairs with 2 % errors and more mutations (rescue / NW paths)
    names_p, p1, p2 = synth.simulate_pairs(genome, 1200, seed=102, err=0.02, mut=0.004, indel_frac=0.3)
    fp = os.path.join(OUT, "pe_interleaved.fq")
    with open(fp, "wb") as fh:
        for i, nm in enumerate(names_p):
            for mate, arr in ((1, p1), (2, p2)):
                fh.write(b"@" + nm.encode() + b"\t/%d\n" % mate + arr[i].tobytes() + b"\n+\n" + b"5" * arr.shape[1] + b"\n")
    case("pe_interleaved", ["-f", fp, "-p"])
    # single-end FASTQ and FASTA
    fs = os.path.join(OUT, "se.fq")
    synth.write_fastq(fs, names[:3000], r1[:3000])
    case("se", ["-f", fs])
    case("se_m", ["-f", fs, "-m"])
    fa = os.path.join(OUT, "se.fa")
    synth.write_fasta_reads(fa, names_p[:500], p1[:500])
    case("se_fasta", ["-f", fa])
    # -g 2 (MaxGaps)
    case("pe_g2", ["-f", f1, "-f2", f2, "-g", "2"])
    # long reads, -pacbio: 60 reads of 2.5 kb with 10 % substitutions + 2 % indel errors
    ln, lr = synth.simulate_long_reads(genome, 60, seed=103, read_len=2500, err=0.12, indel_err_frac=0.2)
    fl = os.path.join(OUT, "pacbio.fq")
    synth.write_fastq(fl, ln, lr)
    case("pacbio", ["-f", fl, "-pacbio"])
    # gz input path
    gz_copy(f1, f1 + ".gz"); gz_copy(f2, f2 + ".gz")
    case("pe_gz", ["-f", f1 + ".gz", "-f2", f2 + ".gz"])
    for f in (f1, f2, fp, fs, fa, fl):
        if not f.endswith(".gz"):
            if not os.path.exists(f + ".gz"):
                gz_copy(f, f + ".gz")
            os.remove(f)
    print("golden SAM written to", OUT)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""tools/probe_odd_inputs_vs_reference.py -- odd but accepted inputs and option combinations, product (or host pipeline on the CPU
oracle backend) against oracle/_ref/kart -t 1: records cut short, lines beyond the gzgets() buffer, entries without '@', empty
reads, missing '+' lines, blank lines, FASTA shapes, several libraries of different formats, missing files, -pacbio with mate files,
-p / -m / -g combinations.  VALIDATION TOOL (needs oracle/_ref).  KART_FUZZ_BIN=<repo>/kart_amd/bin/kart-amd on a GPU box.
Run from a scratch directory; prints one line per case: "same" or what differs.  Cases marked (B-5) have mates of unequal length,
where the reference reads uninitialised bytes -- a difference there is expected."""
import gzip
import os
import subprocess
import sys

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from kart_amd import synth  # noqa: E402
from kart_amd.index_build import read_fasta  # noqa: E402

AMD = os.environ.get("KART_FUZZ_BIN", R + "/tests/_build/kart-host-oracle")
REF = R + "/oracle/_ref/kart"
SMALL = R + "/tests/golden/idx/small"
SAM = R + "/tests/golden/sam/"
bad = 0


def run(binary, args, t, env=None):
    out = "probe_out.sam"
    if os.path.exists(out):
        os.remove(out)
    r = subprocess.run([binary, "-silent", "-i", SMALL] + args + ["-t", t, "-o", out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       env=dict(os.environ, **(env or {})))
    return (open(out, "rb").read() if os.path.exists(out) else None), r.returncode


def cmp(tag, args, envs=({}, {"KART_AMD_NO_MMAP": "1"})):
    global bad
    want, rc_ref = run(REF, args, "1")
    res = []
    for env in envs:
        got, rc = run(AMD, args, "4", env)
        if rc_ref != 0 and rc_ref not in (1,):
            res.append("reference died (%d), product status %d" % (rc_ref, rc))
        elif got == want and (rc == rc_ref):
            res.append("same")
        else:
            if "(B-5)" not in tag:
                bad += 1
            x, y = (got or b"").split(b"\n"), (want or b"").split(b"\n")
            nd = [i for i, (p, q) in enumerate(zip(x, y)) if p != q]
            res.append("DIFF status %d/%d, %d/%d lines, %d differing, first %s" % (rc, rc_ref, len(x), len(y), len(nd), nd[:2]))
    print("%-52s %s" % (tag, " | ".join(res)), flush=True)


def put(name, lines, gz=False):
    data = b"\n".join(lines) + b"\n"
    open(name, "wb").write(gzip.compress(data) if gz else data)
    return name


def main():
    genome = {n: s for n, _, s in read_fasta(R + "/tests/golden/small.fa")}
    r1 = gzip.open(SAM + "pe_1.fq.gz").read().split(b"\n")[:4 * 1200]
    r2 = gzip.open(SAM + "pe_2.fq.gz").read().split(b"\n")[:4 * 1200]
    at = 4 * 300
    for gz in (False, True):
        e, t = (".fq.gz", "gz ") if gz else (".fq", "plain ")
        whole2 = put("w2" + e, r2, gz)
        for name, lines in (("after header", r1[:at + 1]), ("mid sequence", r1[:at + 1] + [r1[at + 1][:70]]), ("after sequence", r1[:at + 2]),
                            ("after +", r1[:at + 3]), ("mid qualities", r1[:at + 3] + [r1[at + 3][:70]])):
            data = b"\n".join(lines) + (b"" if name.startswith("mid") else b"\n")
            open("c1" + e, "wb").write(gzip.compress(data) if gz else data)
            cmp(t + "cut " + name + ", single-end", ["-f", "c1" + e])
            cmp(t + "cut " + name + ", mate 1" + (" (B-5)" if gz and name == "after header" else ""), ["-f", "c1" + e, "-f2", whole2])
        a = list(r1); a[at] = b"@"
        cmp(t + 'header "@" only', ["-f", put("d" + e, a, gz)])
        a = list(r1); a[at + 1] = b""; a[at + 3] = b""
        cmp(t + "empty read, single-end", ["-f", put("m" + e, a, gz)])
        cmp(t + "empty read in mate 1", ["-f", "m" + e, "-f2", whole2])
        a = list(r1); del a[at + 2]
        cmp(t + "a record without its + line", ["-f", put("f" + e, a, gz)])
        a = list(r1); a[43] = b"@" + a[43][1:]; a[42] = b"+" + a[40][1:]
        cmp(t + "qualities starting with @, long + line", ["-f", put("q" + e, a, gz)])
        cmp(t + "two paired libraries", ["-f", put("h1" + e, r1[:2800], gz), put("h3" + e, r1[2800:], gz), "-f2", put("h2" + e, r2[:2800], gz), put("h4" + e, r2[2800:], gz)])
        cmp(t + "two blank lines at the end", ["-f", put("be" + e, r1 + [b"", b""], gz)])
        a = list(r1); a[at] = b"X" + a[at][1:]
        cmp(t + "an entry without @", ["-f", put("nh" + e, a, gz)])
    names, reads = synth.simulate_long_reads(genome, 40, seed=3, read_len=1200, err=0.02)
    synth.write_fastq("l12.fq", names, reads)
    cmp("gz 1200-base reads without -pacbio", ["-f", put("l12.fq.gz", open("l12.fq", "rb").read().split(b"\n")[:-1], True)])
    synth.write_fastq("lh.fq", [n + " " + "x" * 1500 for n in names], reads)
    cmp("gz 1500-character headers", ["-f", put("lh.fq.gz", open("lh.fq", "rb").read().split(b"\n")[:-1], True)])
    cmp("plain 1500-character headers", ["-f", "lh.fq"])
    # FASTA shapes
    names, s1, s2 = synth.simulate_pairs(genome, 800, seed=41, err=0.02, mut=0.002, indel_frac=0.3)

    def fasta(path, reads, width=None, gz=False, mate=None, blank_every=0, empty_at=None):
        out = []
        for i, (n, r) in enumerate(zip(names, reads)):
            s = r.tobytes()
            out.append(b">" + n.encode() + (b"/%d" % mate if mate else b""))
            if empty_at == i:
                s = b""
            out += ([s[k:k + width] for k in range(0, len(s), width)] or [b""]) if width else [s]
            if blank_every and i % blank_every == 3:
                out.append(b"")
        return put(path, out, gz)

    cmp("fasta pairs, one line per read", ["-f", fasta("a1.fa", s1, mate=1), "-f2", fasta("a2.fa", s2, mate=2)])
    cmp("fasta pairs, 60 columns", ["-f", fasta("b1.fa", s1, 60, mate=1), "-f2", fasta("b2.fa", s2, 60, mate=2)])
    cmp("gz fasta pairs", ["-f", fasta("c1.fa.gz", s1, gz=True, mate=1), "-f2", fasta("c2.fa.gz", s2, gz=True, mate=2)])
    cmp("fasta with blank lines", ["-f", fasta("d1.fa", s1, 60, blank_every=9)])
    cmp("fasta with an empty record", ["-f", fasta("e1.fa", s1, 60, empty_at=100)])
    cmp("gz fasta with an empty record", ["-f", fasta("e1.fa.gz", s1, gz=True, empty_at=100)])
    cmp("fasta -m", ["-f", "a1.fa", "-m"])
    synth.write_fastq("g1.fq", names, s1, mate=1)
    synth.write_fastq("g2.fq", names, s2, mate=2)
    for f in ("g1", "g2"):
        open(f + ".fq.gz", "wb").write(gzip.compress(open(f + ".fq", "rb").read()))
    cmp("gz pairs -m -g 3", ["-f", "g1.fq.gz", "-f2", "g2.fq.gz", "-m", "-g", "3"])
    cmp("pairs -g 0", ["-f", "g1.fq", "-f2", "g2.fq", "-g", "0"])
    cmp("short reads -pacbio", ["-f", "g1.fq.gz", "-pacbio"])
    cmp("short pairs -pacbio", ["-f", "g1.fq", "-f2", "g2.fq", "-pacbio"])
    cmp("libraries: fastq pairs + fasta pairs", ["-f", "g1.fq", "a1.fa", "-f2", "g2.fq", "a2.fa"])
    cmp("libraries: fasta pairs + gz fastq pairs", ["-f", "a1.fa", "g1.fq.gz", "-f2", "a2.fa", "g2.fq.gz"])
    cmp("a pair of files of different formats", ["-f", "g1.fq", "-f2", "a2.fa"])
    cmp("... followed by a good library", ["-f", "g1.fq", "g1.fq", "-f2", "a2.fa", "g2.fq"])
    cmp("a missing -f file", ["-f", "nonexistent.fq"])
    cmp("a missing -f2 file", ["-f", "g1.fq", "-f2", "nonexistent.fq"])
    cmp("first library missing, second fine", ["-f", "nonexistent.fq", "g1.fq"])
    cmp("three single-end libraries of three formats", ["-f", "g1.fq", "a1.fa", "g2.fq.gz"])
    cmp("an empty input", ["-f", "/dev/null"])
    # long reads with mate files
    ln, l1 = synth.simulate_long_reads(genome, 61, seed=51, read_len=1500, err=0.12, indel_err_frac=0.3)
    _, l2 = synth.simulate_long_reads(genome, 61, seed=52, read_len=1800, err=0.12, indel_err_frac=0.3)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    rng = np.random.default_rng(1)
    l2[5], l1[7] = rng.choice(acgt, 1800), rng.choice(acgt, 1500)
    synth.write_fastq("x1.fq", ln, l1)
    synth.write_fastq("x2.fq", ln, l2)
    synth.write_fastq("xi.fq", [n for n in ln[:60] for _ in (0, 1)], [x for p in zip(l1[:60], l2[:60]) for x in p])
    cmp("-pacbio with mate files (61 reads each)", ["-f", "x1.fq", "-f2", "x2.fq", "-pacbio"])
    cmp("-pacbio -p", ["-f", "xi.fq", "-p", "-pacbio"])
    cmp("-pacbio with mate files -m", ["-f", "x1.fq", "-f2", "x2.fq", "-pacbio", "-m"])
    cmp("pairs of 1500 / 1800 bases without -pacbio (B-5)", ["-f", "x1.fq", "-f2", "x2.fq"])
    print("unexpected differences:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

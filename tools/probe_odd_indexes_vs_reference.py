#!/usr/bin/env python3
"""tools/probe_odd_indexes_vs_reference.py -- odd reference FASTA files (lower case, IUPAC codes, N runs, an N-only contig, contigs
of 1 and 7 bases, 60 contigs, CRLF, blank lines, no final newline, tabs in names): (1) the index writer kart_amd.index_build against
the reference's bwt_index, five files byte for byte; (2) 600 pairs drawn from ANY window of the concatenated text -- contig
boundaries and N runs included -- mapped by the product (KART_FUZZ_BIN, default the host pipeline on the CPU oracle backend) and by
oracle/_ref/kart -t 1, paired / single-end / -m.  VALIDATION TOOL (needs oracle/_ref).  Run from a scratch directory."""
import filecmp
import os
import random
import subprocess
import sys

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from kart_amd import index_build, synth  # noqa: E402

AMD = os.environ.get("KART_FUZZ_BIN", R + "/tests/_build/kart-host-oracle")
REF, REFIX = R + "/oracle/_ref/kart", R + "/oracle/_ref/bwt_index"


def run(binary, idx, args, t):
    out = "ix_out.sam"
    r = subprocess.run([binary, "-silent", "-i", idx] + args + ["-t", t, "-o", out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    return (open(out, "rb").read() if r.returncode == 0 else None), r.returncode


def main():
    random.seed(7)

    def rnd(n):
        return "".join(random.choice("ACGT") for _ in range(n))

    cases = {
        "lowercase": ">c1\n" + rnd(3000).lower() + "\n>c2\n" + rnd(2000) + "\n",
        "iupac": ">c1\n" + rnd(1500) + "RYKMSWBDHVN" * 20 + rnd(1500) + "\n>c2 desc here\n" + rnd(2500) + "\n",
        "n_runs": ">c1\n" + rnd(1000) + "N" * 500 + rnd(1000) + "\n>c2\n" + "N" * 300 + rnd(2000) + "N" * 200 + "\n",
        "multiline60": ">c1\n" + "\n".join(rnd(60) for _ in range(50)) + "\n>c2\twith tab\n" + "\n".join(rnd(60) for _ in range(40)) + "\n",
        "no_final_nl": ">c1\n" + rnd(3000) + "\n>c2\n" + rnd(2000),
        "crlf": ">c1\r\n" + rnd(3000) + "\r\n>c2\r\n" + rnd(2000) + "\r\n",
        "blank_lines": ">c1\n" + rnd(1500) + "\n\n" + rnd(1500) + "\n\n>c2\n" + rnd(2000) + "\n",
        "tiny_contigs": ">a\nACGTACG\n>b\n" + rnd(3000) + "\n>c\nA\n>d\n" + rnd(2500) + "\n",
        "n_only_contig": ">a\n" + rnd(3000) + "\n>nn\n" + "N" * 400 + "\n>c\n" + rnd(2500) + "\n",
        "many_contigs": "".join(">k%d\n%s\n" % (i, rnd(random.randint(200, 900))) for i in range(60)),
        "odd_length": ">c1\n" + rnd(3001) + "\n>c2\n" + rnd(1999) + "\n",
    }
    bad = 0
    rng = np.random.default_rng(3)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    for name, fa in cases.items():
        open(name + ".fa", "w", newline="").write(fa)
        if subprocess.run([REFIX, name + ".fa", "ref_" + name], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL).returncode != 0:
            print(name, "reference bwt_index failed")
            continue
        index_build.build_index(name + ".fa", "our_" + name, device="cpu")
        files = [ext + (":same" if filecmp.cmp("ref_%s.%s" % (name, ext), "our_%s.%s" % (name, ext), shallow=False) else ":DIFF") for ext in ("amb", "ann", "bwt", "pac", "sa")]
        bad += sum(f.endswith("DIFF") for f in files)
        whole = np.concatenate([np.frombuffer(np.asarray(s).tobytes().upper(), np.uint8) for _, _, s in index_build.read_fasta(name + ".fa")])
        names, r1, r2 = [], [], []
        for i in range(600):
            ins = int(rng.integers(200, 500))
            p = int(rng.integers(0, max(1, len(whole) - ins)))
            frag = whole[p:p + ins].copy()
            if len(frag) < 150:
                continue
            a, b = frag[:150].copy(), synth.revcomp(frag[-150:].copy())
            for x in (a, b):
                e = rng.random(150) < 0.02
                x[e] = acgt[rng.integers(0, 4, int(e.sum()))]
            if rng.random() < 0.5:
                a, b = b, a
            names.append("q%d_%d" % (i, p)); r1.append(a); r2.append(b)
        synth.write_fastq("q1.fq", names, r1, mate=1)
        synth.write_fastq("q2.fq", names, r2, mate=2)
        res = []
        for tag, args in (("pe", ["-f", "q1.fq", "-f2", "q2.fq"]), ("se", ["-f", "q1.fq"]), ("pe -m", ["-f", "q1.fq", "-f2", "q2.fq", "-m"])):
            want, _ = run(REF, "ref_" + name, args, "1")
            got, _ = run(AMD, "our_" + name, args, "4")
            if got == want:
                res.append(tag + ": same")
                continue
            x, y = (got or b"").split(b"\n"), (want or b"").split(b"\n")
            # (-m pairs: a FLAG the reference never assigns is heap contents there)
            hard = [i for i, (p_, q_) in enumerate(zip(x, y)) if p_.split(b"\t")[:1] + p_.split(b"\t")[2:] != q_.split(b"\t")[:1] + q_.split(b"\t")[2:]]
            if hard or len(x) != len(y) or tag != "pe -m":
                bad += 1
            res.append(tag + ": DIFF (%d lines beyond FLAG)" % len(hard))
        print("%-14s %s || %s" % (name, " ".join(files), " | ".join(res)), flush=True)
    print("unexpected differences:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

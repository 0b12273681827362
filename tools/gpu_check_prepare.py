#!/usr/bin/env python3
"""tools/gpu_check_prepare.py -- inputs and expected outputs for tools/gpu_check_hostpath.sh, made in the BUILD container (it runs the
unmodified reference, oracle/_ref/kart -t 1, which does not exist on the GPU box): tests/_build/gpucheck/ (git-ignored, travels with
the gpurun snapshot).  Short pairs and long reads with literal '-' / N runs on the small golden index; 3000 x 7 kb reads at 15 % error
on a 5 Mbp two-contig genome with its own index."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kart_amd import index_build, synth  # noqa: E402

D = os.path.join(ROOT, "tests", "_build", "gpucheck")
SMALL = os.path.join(ROOT, "tests", "golden", "idx", "small")
REF = os.path.join(ROOT, "oracle", "_ref", "kart")


def reference(args, out):
    subprocess.run([REF, "-silent", "-t", "1"] + args + ["-o", out], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def main():
    os.makedirs(D, exist_ok=True)
    genome = {n: s for n, _, s in index_build.read_fasta(os.path.join(ROOT, "tests", "golden", "small.fa"))}
    rng = np.random.default_rng(12)

    def spoil(reads):
        out = []
        for i, r in enumerate(reads):
            r = np.array(r, copy=True)
            ch = ord("-") if i % 3 else ord("N")
            if i % 2 == 0:
                r[rng.integers(0, len(r), size=int(rng.integers(1, 6)))] = ch
            if i % 5 == 0:
                p0 = int(rng.integers(0, len(r) - 12))
                r[p0:p0 + int(rng.integers(2, 9))] = ch
            if i % 11 == 0:
                r[:3] = ord("-")
                r[-2:] = ord("-")
            out.append(r)
        return out

    names, r1, r2 = synth.simulate_pairs(genome, 1500, seed=21, err=0.02, mut=0.002, indel_frac=0.3)
    synth.write_fastq(D + "/d_1.fq", names, spoil(r1), mate=1)
    synth.write_fastq(D + "/d_2.fq", names, spoil(r2), mate=2)
    ln, lr = synth.simulate_long_reads(genome, 150, seed=22, read_len=2500, err=0.15, indel_err_frac=0.3)
    synth.write_fastq(D + "/d_long.fq", ln, spoil(lr))
    reference(["-i", SMALL, "-f", D + "/d_1.fq", "-f2", D + "/d_2.fq"], D + "/d_short.ref.sam")
    reference(["-i", SMALL, "-f", D + "/d_long.fq", "-pacbio"], D + "/d_long.ref.sam")

    g = synth.make_genome([("chrA", 3_000_000), ("chrB", 2_000_000)], seed=77, repeat_frac=0.05)
    synth.write_fasta(D + "/g.fa", g)
    index_build.build_index(D + "/g.fa", D + "/g", device="cpu")
    names, reads = synth.simulate_long_reads(g, 3000, seed=5, read_len=7000, err=0.15)
    synth.write_fastq(D + "/long.fq", names, reads)
    reference(["-i", D + "/g", "-f", D + "/long.fq", "-pacbio"], D + "/long.ref.sam")
    os.remove(D + "/g.fa")


if __name__ == "__main__":
    main()

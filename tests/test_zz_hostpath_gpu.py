"""GPU (runs last): inputs whose handling lives in the host code around the kernels, through the product binary -- bgzip-ped
FASTQ inflated member by member, reads with literal '-' / N runs (the CIGAR scan of the reference takes a literal '-' for a gap
column: such fragments stay with the host, pass 2 of everything else is read off the device's op strings).  The same inputs as
tools/gpu_check_hostpath.sh / gpu_check_bgzf.sh, which ran on the box in round 4 (profiles/r04zu, r04zx)."""
import gzip
import os
import random
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, SMALL_PREFIX

pytestmark = pytest.mark.gpu
KART_AMD = os.path.join(ROOT, "kart_amd", "bin", "kart-amd")
KART_REF = os.path.join(ROOT, "oracle", "_ref", "kart")
SAM = os.path.join(GOLDEN, "sam")


def run(binary, args, out, env=None):
    r = subprocess.run([binary, "-silent", "-i", SMALL_PREFIX] + args + ["-o", out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stdout.decode()[-400:]
    return open(out, "rb").read()


def test_bgzf_pairs_match_the_golden_sam(built_lib, tmp_path):
    from bgzf_util import bgzf
    rng = random.Random(4)
    files = []
    for m, block in ((1, 0xff00), (2, 3000)):
        raw = gzip.open(os.path.join(SAM, "pe_%d.fq.gz" % m)).read()
        path = str(tmp_path / ("b%d.fq.gz" % m))
        open(path, "wb").write(bgzf(raw, block, rng if m == 2 else None))
        files.append(path)
    want = gzip.open(os.path.join(SAM, "pe.sam.gz")).read()
    out = str(tmp_path / "o.sam")
    assert run(KART_AMD, ["-f", files[0], "-f2", files[1], "-t", "16"], out) == want
    assert run(KART_AMD, ["-f", files[0], "-f2", files[1], "-t", "16"], out, {"KART_AMD_NO_BGZF": "1"}) == want


def test_reads_with_dashes_match_live_reference(built_lib, tmp_path):
    assert os.path.exists(KART_REF), "oracle/_ref/kart did not travel to the GPU box"
    from kart_amd import synth
    from kart_amd.index_build import read_fasta
    genome = {n: s for n, _, s in read_fasta(os.path.join(GOLDEN, "small.fa"))}
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
    f1, f2, fl = (str(tmp_path / n) for n in ("d_1.fq", "d_2.fq", "d_long.fq"))
    synth.write_fastq(f1, names, spoil(r1), mate=1)
    synth.write_fastq(f2, names, spoil(r2), mate=2)
    ln, lr = synth.simulate_long_reads(genome, 150, seed=22, read_len=2500, err=0.15, indel_err_frac=0.3)
    synth.write_fastq(fl, ln, spoil(lr))
    out = str(tmp_path / "o.sam")
    short = ["-f", f1, "-f2", f2]
    assert run(KART_AMD, short + ["-t", "16"], out) == run(KART_REF, short + ["-t", "1"], str(tmp_path / "r.sam"))
    long_ = ["-f", fl, "-pacbio"]
    want = run(KART_REF, long_ + ["-t", "1"], str(tmp_path / "r.sam"))
    assert run(KART_AMD, long_ + ["-t", "16"], out) == want
    assert run(KART_AMD, long_ + ["-t", "16"], out, {"KART_AMD_FINISH_STRINGS": "1"}) == want

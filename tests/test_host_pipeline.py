"""CPU: the product's HOST pipeline (kart_amd/csrc/host: FASTQ/FASTA(.gz) reader, chaining, pairing with
the EstDistance feedback, rescue, two-pass report around a batched NW call, flags, MAPQ, SAM text) bound
to the CPU oracle backend (tests/cpu_backend) must reproduce the reference's golden SAM byte for byte.
The golden files were written by the unmodified reference binary (oracle/make_golden_sam.py)."""
import gzip
import os
import subprocess

import pytest

from conftest import GOLDEN, ROOT, SMALL_PREFIX

SAM = os.path.join(GOLDEN, "sam")
CASES = {
    "se": ["-f", "se.fq.gz"],
    "se_m": ["-f", "se.fq.gz", "-m"],
    "se_fasta": ["-f", "se.fa"],                       # plain FASTA path (gz FASTA is single-line only in the reference)
    "pe": ["-f", "pe_1.fq.gz", "-f2", "pe_2.fq.gz"],  # 2.25 chunks: EstDistance switches from 1500 to the estimate
    "pe_plain": ["-f", "pe_1.fq", "-f2", "pe_2.fq"],  # getline() reader instead of gzgets()
    "pe_m": ["-f", "pe_1.fq.gz", "-f2", "pe_2.fq.gz", "-m"],
    "pe_g2": ["-f", "pe_1.fq.gz", "-f2", "pe_2.fq.gz", "-g", "2"],
    "pe_interleaved": ["-f", "pe_interleaved.fq", "-p"],
    "pacbio": ["-f", "pacbio.fq.gz", "-pacbio"],
}
GOLD_OF = {"pe_plain": "pe"}


def materialise(tmp, name):
    """fixtures are stored gzipped; the plain-file cases get an unpacked copy"""
    src = os.path.join(SAM, name)
    if os.path.exists(src):
        return src
    dst = os.path.join(tmp, name)
    if not os.path.exists(dst):
        with gzip.open(src + ".gz") as fi, open(dst, "wb") as fo:
            fo.write(fi.read())
    return dst


def run_case(binary, case, tmp):
    args = [materialise(tmp, a) if (a.endswith((".fq", ".fa", ".gz"))) else a for a in CASES[case]]
    out = os.path.join(tmp, case + ".sam")
    r = subprocess.run([binary, "-silent", "-i", SMALL_PREFIX] + args + ["-o", out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()[-500:]
    got = open(out, "rb").read()
    want = gzip.open(os.path.join(SAM, GOLD_OF.get(case, case) + ".sam.gz")).read()
    return got, want, r.stdout.decode()


@pytest.fixture(scope="module")
def host_oracle_binary():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpu_backend")], stdout=subprocess.DEVNULL)
    return os.path.join(ROOT, "tests", "_build", "kart-host-oracle")


@pytest.mark.parametrize("case", sorted(CASES))
def test_host_pipeline_matches_reference_sam(case, host_oracle_binary, tmp_path):
    got, want, _ = run_case(host_oracle_binary, case, str(tmp_path))
    assert got == want


def test_summary_statistics(host_oracle_binary, tmp_path):
    _, _, log = run_case(host_oracle_binary, "pe", str(tmp_path))
    assert "All the 9000 paired-end reads have been processed" in log
    assert "# of total mapped sequences" in log and "average insert size" in log

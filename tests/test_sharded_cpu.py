"""CPU: one process per device over contiguous chunk ranges of one library (kart_amd/csrc/host/detail/shard.inc) -- the
multi-GPU form of the reference's chunk fan-out (reference src/Mapping.cpp:504-512, 716-717) including its EstDistance
feedback (:533-540, :209-213).  The host pipeline bound to the CPU oracle backend, `-gpu 0,1[,..]` = N processes: the SAM
must be byte-identical to the single-process output, i.e. to the reference's `-t 1`."""
import gzip
import os
import subprocess

import pytest

from conftest import GOLDEN, ROOT, SMALL_PREFIX
from test_host_pipeline import CASES, host_oracle_binary, materialise, run_case   # noqa: F401  (fixture)


@pytest.mark.parametrize("devices", ["0,1", "0,1,2", "0,0,0,0,0"])
@pytest.mark.parametrize("case", ["pe_plain", "pe_interleaved", "edge_se", "pe", "se_fasta", "edge_multi_lib"])
def test_sharded_golden(case, devices, host_oracle_binary, tmp_path):
    """golden SAMs of the reference: plain FASTQ is split (2.25 chunks: the last shards are empty), gz / FASTA / several
    libraries are mapped by shard 0 alone -- same bytes either way"""
    got, want, log = run_case(host_oracle_binary, case, str(tmp_path), ["-gpu", devices, "-t", "6"])
    assert got == want
    if case == "pe_plain":
        assert "All the 9000 paired-end reads have been processed" in log and "# of paired sequences = 8994" in log


@pytest.fixture(scope="module")
def moving_estimate_input(tmp_path_factory):
    """40 chunks whose insert sizes drift (300 -> 180): the estimate keeps moving below MaxInsertSize, so rescue windows and
    pairing tests depend on the totals in front of every chunk -- the case the settle step of a later shard exists for"""
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "kart")
    if not os.path.exists(ref_bin):
        pytest.skip("oracle/_ref/kart not present")
    from kart_amd import synth
    from kart_amd.index_build import read_fasta
    tmp = tmp_path_factory.mktemp("moving")
    genome = {n: s for n, _, s in read_fasta(os.path.join(GOLDEN, "small.fa"))}
    f1, f2 = str(tmp / "s_1.fq"), str(tmp / "s_2.fq")
    with open(f1, "wb") as o1, open(f2, "wb") as o2:
        for part, ins in enumerate((300, 260, 220, 180)):
            names, r1, r2 = synth.simulate_pairs(genome, 20000, seed=950 + part, err=0.02, mut=0.002, indel_frac=0.3, ins_mean=float(ins), ins_sd=ins / 8.0)
            names = ["p%d_%s" % (part, n) for n in names]
            p1, p2 = str(tmp / "t1.fq"), str(tmp / "t2.fq")
            synth.write_fastq(p1, names, r1, mate=1)
            synth.write_fastq(p2, names, r2, mate=2)
            o1.write(open(p1, "rb").read())
            o2.write(open(p2, "rb").read())
    ref = str(tmp / "ref.sam")
    subprocess.run([ref_bin, "-silent", "-i", SMALL_PREFIX, "-f", f1, "-f2", f2, "-o", ref, "-t", "1"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return f1, f2, open(ref, "rb").read()


@pytest.mark.parametrize("devices", ["0", "0,1", "0,1,2", "0,1,2,3,4,5,6,7"])
def test_sharded_moving_estimate_matches_live_reference(devices, moving_estimate_input, host_oracle_binary, tmp_path):
    f1, f2, ref = moving_estimate_input
    out = str(tmp_path / "o.sam")
    r = subprocess.run([host_oracle_binary, "-silent", "-i", SMALL_PREFIX, "-f", f1, "-f2", f2, "-o", out, "-gpu", devices, "-t", "8"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=dict(os.environ, KART_AMD_VERBOSE="1"))
    assert r.returncode == 0, r.stdout.decode()[-600:]
    assert open(out, "rb").read() == ref
    assert "All the 160000 paired-end reads" in r.stdout.decode()


def test_a_failing_shard_ends_the_run(host_oracle_binary, tmp_path):
    """a shard that cannot load its index must not leave the others waiting"""
    fq = materialise(str(tmp_path), "pe_1.fq")
    bad = str(tmp_path / "nonexistent")
    for e in (".ann", ".amb", ".pac"):
        open(bad + e, "w").write("")
    r = subprocess.run([host_oracle_binary, "-silent", "-i", bad, "-f", fq, "-o", str(tmp_path / "o.sam"), "-gpu", "0,1"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120)
    assert r.returncode != 0


@pytest.mark.parametrize("devices", ["0,1", "0,1,2,3"])
@pytest.mark.parametrize("case", ["pe_plain", "edge_se", "pe"])
def test_sharded_parts_concatenate_to_the_golden_file(case, devices, host_oracle_binary, tmp_path):
    """-parts: one output file per shard (<out>.<r>); their concatenation is the single-process file byte for byte"""
    args = [materialise(str(tmp_path), a) if a.endswith((".fq", ".fa", ".gz")) else a for a in CASES[case]]
    out = str(tmp_path / "o.sam")
    r = subprocess.run([host_oracle_binary, "-silent", "-i", SMALL_PREFIX] + args + ["-o", out, "-gpu", devices, "-parts", "-t", "6"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()[-500:]
    n = len(devices.split(","))
    got = b"".join(open("%s.%d" % (out, q), "rb").read() for q in range(n))
    from test_host_pipeline import GOLD_OF, SAM
    assert got == gzip.open(os.path.join(SAM, GOLD_OF.get(case, case) + ".sam.gz")).read()


@pytest.mark.parametrize("devices", ["0,1", "0,1,2,3", "0,1,2,3,4,5,6,7"])
def test_sharded_parts_written_while_mapping_match_live_reference(devices, moving_estimate_input, host_oracle_binary, tmp_path):
    """-parts: a later shard writes its text while it maps (speculating EstDistance from rank 0's running totals plus its own) and
    writes the tail of its part again when settling changed a chunk -- the drifting insert size forces exactly that.  The parts'
    concatenation is the reference's -t 1 file either way, and the same with the text held back until the end."""
    f1, f2, ref = moving_estimate_input
    n = len(devices.split(","))
    rewritten = 0
    for env in ({}, {"KART_AMD_NO_EAGER_PARTS": "1"}):
        out = str(tmp_path / ("o%d.sam" % len(env)))
        r = subprocess.run([host_oracle_binary, "-silent", "-i", SMALL_PREFIX, "-f", f1, "-f2", f2, "-o", out, "-gpu", devices, "-parts", "-t", "8"],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=dict(os.environ, KART_AMD_VERBOSE="1", **env))
        assert r.returncode == 0, r.stdout.decode()[-600:]
        assert b"".join(open("%s.%d" % (out, q), "rb").read() for q in range(n)) == ref, env
        if not env:
            for l in r.stdout.decode().splitlines():
                if l.startswith("shard ") and "chunks of text written again" in l:
                    rewritten += int(l.split("chunks mapped again, ")[1].split()[0])
    assert rewritten > 0, "the drifting estimate should have changed at least one chunk of a later shard"

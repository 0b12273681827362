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
    # edge cases (oracle/make_golden_edge.py): read lengths 5..300 in one file, N-rich / lowercase / IUPAC / all-N
    # reads, homopolymers, unmappable reads, cross-contig and far-apart pairs, names with spaces
    "edge_pe": ["-f", "edge_1.fq", "-f2", "edge_2.fq"],
    "edge_se": ["-f", "edge_1.fq"],
    "edge_se_m": ["-f", "edge_2.fq", "-m"],
    "edge_multi_lib": ["-f", "edge_1.fq", "edge_2.fq"],          # two single-end libraries
    # configs[0]: the read files of the reference's own test (run_test.sh:25-27; oracle/make_golden_reftest.py) on the small index
    "ref_test": ["-f", "ref_test_r1.fq", "-f2", "ref_test_r2.fq"],
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


def run_case(binary, case, tmp, extra=()):
    args = [materialise(tmp, a) if (a.endswith((".fq", ".fa", ".gz"))) else a for a in CASES[case]]
    out = os.path.join(tmp, case + ".sam")
    r = subprocess.run([binary, "-silent", "-i", SMALL_PREFIX] + args + list(extra) + ["-o", out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()[-500:]
    got = open(out, "rb").read()
    want = gzip.open(os.path.join(SAM, GOLD_OF.get(case, case) + ".sam.gz")).read()
    return got, want, r.stdout.decode()


def run_case_env(binary, case, tmp, extra, env):
    args = [materialise(tmp, a) if (a.endswith((".fq", ".fa", ".gz"))) else a for a in CASES[case]]
    out = os.path.join(tmp, case + ".sam")
    r = subprocess.run([binary, "-silent", "-i", SMALL_PREFIX] + args + list(extra) + ["-o", out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=dict(os.environ, **env))
    assert r.returncode == 0, r.stdout.decode()[-500:]
    return open(out, "rb").read(), gzip.open(os.path.join(SAM, GOLD_OF.get(case, case) + ".sam.gz")).read(), r.stdout.decode()


UNSET_FLAG = 1 << 20     # not a SAM flag: no assigned value can collide with it


def assert_same_sam_up_to_unset_flags(ref: bytes, got: bytes) -> int:
    """Every line identical, except that a record whose FLAG the reference never assigns (heap contents there,
    SURVEY.md App. B-12: a printed candidate other than iBestAlnCanIdx of a mate with score > sub_score in -m
    paired mode) -- which the product marks with UNSET_FLAG -- is compared on all other columns only."""
    la, lb = ref.split(b"\n"), got.split(b"\n")
    assert len(la) == len(lb)
    masked = 0
    for x, y in zip(la, lb):
        if x == y:
            continue
        fx, fy = x.split(b"\t"), y.split(b"\t")
        assert len(fy) > 1 and int(fy[1]) == UNSET_FLAG, (x[:120], y[:120])      # any other difference is a real one
        assert fx[:1] + fx[2:] == fy[:1] + fy[2:], (x[:120], y[:120])
        masked += 1
    return masked


def reference_sam_and_never_assigned_flags(ref_bin, args, tmp):
    """The reference's -t 1 SAM, and -- derived from the reference alone -- the lines whose FLAG it never assigns: the records
    whose FLAG column differs between two runs under MALLOC_PERTURB_=85 and =170 (AlnReportArr is new-ed without initialising
    SamFlag, src/AlignmentCandidates.cpp:636-640; glibc fills every block malloc hands out with the perturbation byte -- with the
    thread cache switched off, whose fast path skips the fill).  Every other column of every line must agree between the two runs."""
    outs = []
    for perturb in (85, 170):
        out = os.path.join(tmp, "ref_%d.sam" % perturb)
        subprocess.run([ref_bin, "-silent", "-t", "1"] + list(args) + ["-o", out], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                       env=dict(os.environ, MALLOC_PERTURB_=str(perturb), GLIBC_TUNABLES="glibc.malloc.tcache_count=0"))
        outs.append(open(out, "rb").read().split(b"\n"))
    a, b = outs
    assert len(a) == len(b)
    never = set()
    for i, (x, y) in enumerate(zip(a, b)):
        if x != y:
            fx, fy = x.split(b"\t"), y.split(b"\t")
            assert fx[:1] + fx[2:] == fy[:1] + fy[2:], (x[:120], y[:120])      # only the FLAG may depend on the heap
            never.add(i)
    return a, never


def assert_sam_equals_reference_with_its_own_mask(ref_lines, never, got: bytes) -> int:
    """every line of `got` equals the reference's, except that exactly the records in `never` (the reference's never-assigned
    FLAGs, found WITHOUT looking at the product) carry UNSET_FLAG in the FLAG column: the product's sentinel set must BE that set"""
    lb = got.split(b"\n")
    assert len(ref_lines) == len(lb)
    sentinel = set()
    for i, (x, y) in enumerate(zip(ref_lines, lb)):
        fy = y.split(b"\t")
        if len(fy) > 1 and not y.startswith(b"@") and fy[1] == str(UNSET_FLAG).encode():
            sentinel.add(i)
            fx = x.split(b"\t")
            assert fx[:1] + fx[2:] == fy[:1] + fy[2:], (x[:120], y[:120])
        else:
            assert x == y or i in never, (i, x[:120], y[:120])
    assert sentinel == never, (sorted(sentinel ^ never)[:10], len(sentinel), len(never))
    return len(never)


@pytest.fixture(scope="module")
def host_oracle_binary():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpu_backend")], stdout=subprocess.DEVNULL)
    return os.path.join(ROOT, "tests", "_build", "kart-host-oracle")


@pytest.mark.parametrize("case", sorted(CASES))
def test_host_pipeline_matches_reference_sam(case, host_oracle_binary, tmp_path):
    got, want, _ = run_case(host_oracle_binary, case, str(tmp_path))
    assert got == want


@pytest.mark.parametrize("back_every", [1, 3])
def test_pacbio_with_the_fragment_service(back_every, host_oracle_binary, tmp_path):
    """-pacbio as the product runs it: the workers plan a batch's reads, every fragment pair goes through ONE batched
    GenerateNormalPairAlignment call (kg_fragments_batch there; here the oracle's restatement behind the same backend interface),
    the workers stitch the results.  back_every 3: a third of the jobs come back with status != 0 (outside the kernels'
    envelope) and are planned by the host."""
    got, want, log = run_case_env(host_oracle_binary, "pacbio", str(tmp_path), ["-t", "4"],
                                  {"KART_ORACLE_FRAGMENTS": str(back_every), "KART_AMD_VERBOSE": "1"})
    assert got == want
    line = [ln for ln in log.splitlines() if ln.startswith("fragment pairs")][0]
    sent, back = (int(x) for x in __import__("re").findall(r": (\d+)", line))
    assert sent > 0 and (back == 0 if back_every == 1 else 0 < back < sent)


def test_pass2_from_op_strings_equals_pass2_from_gapped_strings():
    """add_cigar_ops / local_quality_ok_ops / finish_head_ops / finish_tail_ops (pass 2 of the long-read report since round 4) against
    add_cigar / local_quality_ok / finish_head / finish_tail (src/tools.cpp:49-104,255-290,314-394 restated on the gapped strings) on
    300 000 random alignments: CIGAR elements, score and the trimmed pair must agree (tests/cpu_backend/ops_selftest.cpp)."""
    target = os.path.join("..", "..", "tests", "_build", "ops_selftest")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpu_backend"), target], stdout=subprocess.DEVNULL)
    for seed in ("1", "2"):
        r = subprocess.run([os.path.join(ROOT, "tests", "_build", "ops_selftest"), "150000", seed], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        assert r.returncode == 0 and b"cases identical" in r.stdout, r.stdout.decode()[-400:]


def test_summary_statistics(host_oracle_binary, tmp_path):
    _, _, log = run_case(host_oracle_binary, "pe", str(tmp_path))
    assert "All the 9000 paired-end reads have been processed" in log
    assert "# of total mapped sequences" in log and "average insert size" in log


@pytest.mark.parametrize("ins_mean,threads", [(250, 8), (500, 3)])
def test_speculative_chunks_match_live_reference(ins_mean, threads, host_oracle_binary, tmp_path):
    """30 chunks mapped concurrently under a speculated EstDistance must commit to exactly the -t 1 output
    (short inserts keep the estimate below MaxInsertSize and moving, so rescue windows and pairing tests
    depend on it).  Needs the unmodified reference binary (oracle/_ref/kart), i.e. runs where it was built."""
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "kart")
    if not os.path.exists(ref_bin):
        pytest.skip("oracle/_ref/kart not present")
    from kart_amd import synth
    from kart_amd.index_build import read_fasta
    genome = {n: s for n, _, s in read_fasta(os.path.join(GOLDEN, "small.fa"))}
    names, r1, r2 = synth.simulate_pairs(genome, 60000, seed=900 + ins_mean, err=0.02, mut=0.002, indel_frac=0.3,
                                         ins_mean=float(ins_mean), ins_sd=ins_mean / 8.0)
    f1, f2 = str(tmp_path / "s_1.fq"), str(tmp_path / "s_2.fq")
    synth.write_fastq(f1, names, r1, mate=1)
    synth.write_fastq(f2, names, r2, mate=2)
    outs = []
    for binary, extra in ((ref_bin, ["-t", "1"]), (host_oracle_binary, ["-t", str(threads)])):
        out = str(tmp_path / (os.path.basename(binary) + ".sam"))
        subprocess.run([binary, "-silent", "-i", SMALL_PREFIX, "-f", f1, "-f2", f2, "-o", out] + extra, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        outs.append(open(out, "rb").read())
    assert outs[0] == outs[1]


def test_reads_with_dashes_and_ns_match_live_reference(host_oracle_binary, tmp_path):
    """Characters the reference does not reject: a literal '-' in a read is a base like any other non-ACGT one for seeding and
    nw_alignment, but AddNewCigarElements / the head and tail trimming (src/tools.cpp:49-104,314-394) scan the gapped strings for
    '-' and take it for a gap column; N runs break the 8-mers.  Short pairs and long reads (-pacbio: host planning, the fragment
    service with and without handed-back jobs, pass 2 from the op strings and from the gapped strings) against kart -t 1."""
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "kart")
    if not os.path.exists(ref_bin):
        pytest.skip("oracle/_ref/kart not present")
    import numpy as np
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
                p0 = int(rng.integers(0, len(r) - 12)); r[p0:p0 + int(rng.integers(2, 9))] = ch
            if i % 11 == 0:
                r[:3] = ord("-"); r[-2:] = ord("-")          # at the ends: the head / tail pairs
            out.append(r)
        return out

    names, r1, r2 = synth.simulate_pairs(genome, 1500, seed=21, err=0.02, mut=0.002, indel_frac=0.3)
    f1, f2 = str(tmp_path / "d_1.fq"), str(tmp_path / "d_2.fq")
    synth.write_fastq(f1, names, spoil(r1), mate=1)
    synth.write_fastq(f2, names, spoil(r2), mate=2)
    lnames, lreads = synth.simulate_long_reads(genome, 150, seed=22, read_len=2500, err=0.15, indel_err_frac=0.3)
    fl = str(tmp_path / "d_long.fq")
    synth.write_fastq(fl, lnames, spoil(lreads))

    def run(binary, args, env=None):
        out = str(tmp_path / "o.sam")
        subprocess.run([binary, "-silent", "-i", SMALL_PREFIX] + args + ["-o", out], check=True, stdout=subprocess.DEVNULL,
                       stderr=subprocess.DEVNULL, env=dict(os.environ, **(env or {})))
        return open(out, "rb").read()

    short = ["-f", f1, "-f2", f2]
    assert run(host_oracle_binary, short + ["-t", "3"]) == run(ref_bin, short + ["-t", "1"])
    long_ = ["-f", fl, "-pacbio"]
    want = run(ref_bin, long_ + ["-t", "1"])
    assert want.count(b"\n") > 150
    for env in ({}, {"KART_ORACLE_FRAGMENTS": "1"}, {"KART_ORACLE_FRAGMENTS": "3"}, {"KART_ORACLE_FRAGMENTS": "1", "KART_AMD_FINISH_STRINGS": "1"}):
        assert run(host_oracle_binary, long_ + ["-t", "3"], env) == want, env


def test_records_cut_short_match_live_reference(host_oracle_binary, tmp_path):
    """A FASTQ file that ends inside a record (an interrupted copy): the reference reads the record's remaining lines into one buffer
    without looking at the results (src/GetData.cpp:73, :168-175), so a line that never came is whatever the buffer still held --
    the '+' line (the "qualities" are then "+" and a line break), the bases themselves, or, behind gzgets(), the header line taken
    for the bases.  Every cut position, either mate file, plain and gz, paired and single-end, the mapped-file / inflated-text
    reader and the line reader (KART_AMD_NO_MMAP): byte-identical to kart -t 1.  Not compared: a gz mate-1 file cut inside or right
    after a header in paired mode -- the header text becomes a 24-base mate 1, and the reference encodes mate 2 with mate 1's
    length (src/Mapping.cpp:550, SURVEY App. B-5: uninitialised bytes)."""
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "kart")
    if not os.path.exists(ref_bin):
        pytest.skip("oracle/_ref/kart not present")
    raw = [gzip.open(os.path.join(SAM, "pe_%d.fq.gz" % m)).read().split(b"\n")[:2400] for m in (1, 2)]     # 600 records each
    at = 2000                                                                                              # record 500 starts here

    def variants(lines):
        def upto(n, extra=b""):
            return b"\n".join(lines[:n]) + b"\n" + extra
        return {"after_header": upto(at + 1), "mid_seq": upto(at + 1, lines[at + 1][:70]), "after_seq": upto(at + 2), "mid_plus": upto(at + 2, b"+"),
                "after_plus": upto(at + 3), "mid_qual": upto(at + 3, lines[at + 3][:70]), "mid_header": upto(at, lines[at][:5])}

    def run(binary, args, env=None):
        out = str(tmp_path / "o.sam")
        r = subprocess.run([binary, "-silent", "-i", SMALL_PREFIX] + args + ["-o", out], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                           env=dict(os.environ, **(env or {})))
        return open(out, "rb").read() if r.returncode == 0 else None

    whole = [b"\n".join(x) + b"\n" for x in raw]
    compared = 0
    for gz in (False, True):
        ext = ".fq.gz" if gz else ".fq"
        f1, f2 = str(tmp_path / ("t1" + ext)), str(tmp_path / ("t2" + ext))
        for which in (0, 1):
            for name, data in variants(raw[which]).items():
                for f, d in ((f1, data if which == 0 else whole[0]), (f2, data if which == 1 else whole[1])):
                    open(f, "wb").write(gzip.compress(d) if gz else d)
                modes = [["-f", f1, "-f2", f2]] + ([["-f", f1]] if which == 0 else [])
                for args in modes:
                    if gz and which == 0 and len(args) == 4 and name in ("after_header", "mid_header"):
                        continue                                                                           # App. B-5, see above
                    want = run(ref_bin, args + ["-t", "1"])
                    assert want is not None
                    for env in ({}, {"KART_AMD_NO_MMAP": "1"}):
                        assert run(host_oracle_binary, args + ["-t", "3"], env) == want, (gz, which, name, args[:1], env)
                        compared += 1
    assert compared == 80


def test_gz_text_that_gzgets_reads_differently_matches_live_reference(host_oracle_binary, tmp_path):
    """The reference reads a gz library through gzgets() with a 1000-byte buffer (src/GetData.cpp:152-159): a longer line -- a 1200-base
    read without -pacbio, a 1500-character header -- comes back in pieces that are taken for the record's next lines, and an entry
    whose first line does not start with '@' / '>' ends after that line (:162).  The inflated-text parser hands such a window, and
    the rest of the library, to the line reader (gz_text_regular): first batch or late in a paired library, plain gzip or BGZF."""
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "kart")
    if not os.path.exists(ref_bin):
        pytest.skip("oracle/_ref/kart not present")
    from bgzf_util import bgzf
    from kart_amd import synth
    from kart_amd.index_build import read_fasta
    genome = {n: s for n, _, s in read_fasta(os.path.join(GOLDEN, "small.fa"))}

    def put(name, data):
        path = str(tmp_path / name)
        open(path, "wb").write(data)
        return path

    def same(args):
        outs = []
        for binary, t in ((ref_bin, "1"), (host_oracle_binary, "4")):
            out = str(tmp_path / "o.sam")
            subprocess.run([binary, "-silent", "-i", SMALL_PREFIX] + args + ["-t", t, "-o", out], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            outs.append(open(out, "rb").read())
        return outs[0] == outs[1]

    names, reads = synth.simulate_long_reads(genome, 40, seed=3, read_len=1200, err=0.02)
    plain = str(tmp_path / "l12.fq")
    synth.write_fastq(plain, names, reads)
    assert same(["-f", put("l12.fq.gz", gzip.compress(open(plain, "rb").read()))])
    synth.write_fastq(plain, [n + " " + "x" * 1500 for n in names], reads)
    assert same(["-f", put("lh.fq.gz", gzip.compress(open(plain, "rb").read()))])
    r1 = gzip.open(os.path.join(SAM, "pe_1.fq.gz")).read()
    l2 = gzip.open(os.path.join(SAM, "pe_2.fq.gz")).read().split(b"\n")
    l2[4 * 3000] += b" " + b"y" * 1500                                  # record 3000 of mate file 2: several batches in
    r2 = b"\n".join(l2)
    assert same(["-f", put("p1.fq.gz", gzip.compress(r1)), "-f2", put("p2.fq.gz", gzip.compress(r2))])
    assert same(["-f", put("b1.fq.gz", bgzf(r1)), "-f2", put("b2.fq.gz", bgzf(r2, 7000))])
    l1 = r1.split(b"\n")
    l1[4 * 1000] = b"X" + l1[4 * 1000][1:]
    nh = put("nh.fq.gz", gzip.compress(b"\n".join(l1)))
    assert same(["-f", nh])
    assert same(["-f", nh, "-f2", put("q2.fq.gz", gzip.compress(gzip.open(os.path.join(SAM, "pe_2.fq.gz")).read()))])


def test_pacbio_with_mate_files_matches_live_reference(host_oracle_binary, tmp_path):
    """-pacbio with -f / -f2 (or -p): the reference maps the reads one by one (src/Mapping.cpp:514-529) but prints an even-sized
    chunk through OutputPairedAlignments (:598) -- mate 2 is held reverse-complemented and shown flipped back by a forward report.
    Mate files of 61 reads (the last chunk is odd: printed singly), an unmappable read in either file, gz, -p, -m."""
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "kart")
    if not os.path.exists(ref_bin):
        pytest.skip("oracle/_ref/kart not present")
    import numpy as np
    from kart_amd import synth
    from kart_amd.index_build import read_fasta
    genome = {n: s for n, _, s in read_fasta(os.path.join(GOLDEN, "small.fa"))}
    names, l1 = synth.simulate_long_reads(genome, 61, seed=51, read_len=1500, err=0.12, indel_err_frac=0.3)
    _, l2 = synth.simulate_long_reads(genome, 61, seed=52, read_len=1800, err=0.12, indel_err_frac=0.3)
    rng = np.random.default_rng(1)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    l2[5], l1[7] = rng.choice(acgt, 1800), rng.choice(acgt, 1500)
    f1, f2, fi = (str(tmp_path / n) for n in ("x1.fq", "x2.fq", "xi.fq"))
    synth.write_fastq(f1, names, l1)
    synth.write_fastq(f2, names, l2)
    synth.write_fastq(fi, [n for n in names[:60] for _ in (0, 1)], [x for p in zip(l1[:60], l2[:60]) for x in p])
    g1, g2 = f1 + ".gz", f2 + ".gz"
    for src, dst in ((f1, g1), (f2, g2)):
        open(dst, "wb").write(gzip.compress(open(src, "rb").read()))

    def run(binary, args, env=None):
        out = str(tmp_path / "o.sam")
        subprocess.run([binary, "-silent", "-i", SMALL_PREFIX] + args + ["-o", out], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                       env=dict(os.environ, **(env or {})))
        return open(out, "rb").read()

    for args in (["-f", f1, "-f2", f2, "-pacbio"], ["-f", g1, "-f2", g2, "-pacbio"], ["-f", fi, "-p", "-pacbio"], ["-f", f1, "-f2", f2, "-pacbio", "-m"]):
        want = run(ref_bin, args + ["-t", "1"])
        for env in ({}, {"KART_ORACLE_FRAGMENTS": "1"}):
            assert run(host_oracle_binary, args + ["-t", "4"], env) == want, (args, env)


def test_bgzf_inputs_are_inflated_member_by_member(host_oracle_binary, tmp_path):
    """bgzip-ped FASTQ: the members are inflated side by side (GzText::fill_bgzf) instead of through one gzread() stream; the text --
    and so the SAM -- is what gzgets() reads in the reference.  Full-size and ragged members (records cut anywhere), no EOF member,
    a plain gzip member appended to BGZF ones (the rest goes through gzread()), a BGZF and a plain gz mate file, -p, and the switch
    that turns the path off."""
    import random
    from bgzf_util import bgzf
    r1 = gzip.open(os.path.join(SAM, "pe_1.fq.gz")).read()
    r2 = gzip.open(os.path.join(SAM, "pe_2.fq.gz")).read()
    want = gzip.open(os.path.join(SAM, "pe.sam.gz")).read()
    rng = random.Random(4)

    def put(name, data):
        path = str(tmp_path / name)
        open(path, "wb").write(data)
        return path

    def run(args, env=None):
        out = str(tmp_path / "o.sam")
        r = subprocess.run([host_oracle_binary, "-silent", "-i", SMALL_PREFIX] + args + ["-t", "8", "-o", out], stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, env=dict(os.environ, **(env or {})))
        assert r.returncode == 0, r.stdout.decode()[-400:]
        return open(out, "rb").read()

    b1, b2 = put("b1.fq.gz", bgzf(r1)), put("b2.fq.gz", bgzf(r2))
    assert gzip.open(b1).read() == r1                      # (the writer writes what gzip reads)
    assert run(["-f", b1, "-f2", b2]) == want
    assert run(["-f", put("c1.fq.gz", bgzf(r1, 5000, rng)), "-f2", put("c2.fq.gz", bgzf(r2, 300, rng, eof=False))]) == want
    half = len(r1) // 2
    assert run(["-f", put("d1.fq.gz", bgzf(r1[:half], eof=False) + gzip.compress(r1[half:])), "-f2", b2]) == want
    assert run(["-f", b1, "-f2", put("e2.fq.gz", gzip.compress(r2))]) == want
    assert run(["-f", b1, "-f2", b2], {"KART_AMD_NO_BGZF": "1"}) == want
    inter = gzip.open(os.path.join(SAM, "pe_interleaved.fq.gz")).read()
    assert run(["-f", put("i.fq.gz", bgzf(inter, 20000, rng)), "-p"]) == gzip.open(os.path.join(SAM, "pe_interleaved.sam.gz")).read()


def test_damaged_gz_input_yields_what_the_reference_still_reads(host_oracle_binary, tmp_path):
    """A gz stream with a damaged member: the reference's gzgets() loop sees everything zlib could still inflate before the damage and
    maps it; one large gzread() fails as a whole (rounds 2-3 lost the batch: here the whole library).  GzText::replay() reads the
    file again the way gzgets() does.  Same records as kart -t 1 -- except the one the damage cuts in two, for whose missing lines
    the reference prints stale contents of its line buffer (src/GetData.cpp:160-175)."""
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "kart")
    if not os.path.exists(ref_bin):
        pytest.skip("oracle/_ref/kart not present")
    from bgzf_util import bgzf
    r1 = gzip.open(os.path.join(SAM, "pe_1.fq.gz")).read()
    r2 = gzip.open(os.path.join(SAM, "pe_2.fq.gz")).read()
    good2 = str(tmp_path / "b2.fq.gz")
    open(good2, "wb").write(bgzf(r2))
    for name, data in (("bgzf", bgzf(r1)), ("plain", gzip.compress(r1))):
        d = bytearray(data)
        d[len(d) // 2] ^= 0x55
        bad1 = str(tmp_path / (name + "_1.fq.gz"))
        open(bad1, "wb").write(bytes(d))
        outs = []
        for binary, t in ((ref_bin, "1"), (host_oracle_binary, "8")):
            out = str(tmp_path / "o.sam")
            r = subprocess.run([binary, "-silent", "-i", SMALL_PREFIX, "-f", bad1, "-f2", good2, "-t", t, "-o", out],
                               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            assert r.returncode == 0 or binary == ref_bin         # (where the damage leaves it a negative length the reference aborts)
            outs.append(open(out, "rb").read().split(b"\n") if r.returncode == 0 else None)
        ref, got = outs
        assert len(got) > 1000, (name, len(got))                   # the reads before the damage are mapped
        if ref is None:
            continue
        assert len(got) == len(ref), (name, len(got), len(ref))
        differing = [i for i, (x, y) in enumerate(zip(ref, got)) if x != y]
        assert len(differing) <= 1 and all(i >= len(ref) - 4 for i in differing), (name, differing[:5])


def test_odd_input_files_match_live_reference(host_oracle_binary, tmp_path):
    """inputs the readers must treat exactly like the reference's getline()/gzgets() loops: a second file shorter than the
    first, CRLF line ends, no newline at the end of the file, empty files, an interleaved file with an odd number of records,
    one gz and one plain mate file.  Needs oracle/_ref/kart (live comparison)."""
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "kart")
    if not os.path.exists(ref_bin):
        pytest.skip("oracle/_ref/kart not present")
    src1 = open(materialise(str(tmp_path), "pe_1.fq"), "rb").read().split(b"\n")
    src2 = open(materialise(str(tmp_path), "pe_2.fq"), "rb").read().split(b"\n")

    def make(name, lines, n, eol=b"\n", trail=True, gz=False):
        data = eol.join(lines[: 4 * n]) + (eol if trail else b"")
        path = str(tmp_path / name)
        if gz:
            with gzip.open(path, "wb") as fh:
                fh.write(data)
        else:
            open(path, "wb").write(data)
        return path

    a1, a2 = make("a1.fq", src1, 3000), make("a2.fq", src2, 3000)
    cases = {
        "short_r2": ["-f", a1, "-f2", make("short2.fq", src2, 2500)],
        "crlf": ["-f", make("crlf1.fq", src1, 3000, b"\r\n"), "-f2", make("crlf2.fq", src2, 3000, b"\r\n")],
        "no_final_newline": ["-f", make("nt1.fq", src1, 3000, trail=False), "-f2", make("nt2.fq", src2, 3000, trail=False)],
        "empty": ["-f", make("empty.fq", [], 0, trail=False)],
        "interleaved_odd": ["-f", make("odd.fq", src1, 2999), "-p"],
        "gz_crlf": ["-f", make("crlf1.fq.gz", src1, 3000, b"\r\n", gz=True), "-f2", make("crlf2.fq.gz", src2, 3000, b"\r\n", gz=True)],
        "gz_short_r2": ["-f", make("a1.fq.gz", src1, 3000, gz=True), "-f2", make("short2.fq.gz", src2, 2500, gz=True)],
        "gz_and_plain": ["-f", make("b1.fq.gz", src1, 3000, gz=True), "-f2", a2],
        # several libraries in one run: the pairing statistics carry over from one to the next
        "two_pe_libraries": ["-f", a1, make("c1.fq.gz", src1[12000:], 1500, gz=True), "-f2", a2, make("c2.fq.gz", src2[12000:], 1500, gz=True)],
    }

    def fasta(name, lines, n, wrap=None, lower=False, gz=False):
        rec = []
        for i in range(n):
            seq = lines[4 * i + 1]
            if i % 50 == 7:
                seq = seq[:17] + b"N" * 40 + seq[57:]              # an ambiguity run inside the read
            if lower:
                seq = seq.lower()                                   # ... 'n' is NOT skipped by the 8-mer code, unlike 'N'
            rec.append(b">" + lines[4 * i][1:])
            rec += [seq[a:a + wrap] for a in range(0, len(seq), wrap)] if wrap else [seq]
        return make(name, rec, len(rec), gz=gz)

    cases.update({
        "fasta_pe": ["-f", fasta("s1.fa", src1, 2000), "-f2", fasta("s2.fa", src2, 2000)],
        "fasta_wrapped_pe": ["-f", fasta("w1.fa", src1, 2000, wrap=60), "-f2", fasta("w2.fa", src2, 2000, wrap=60)],
        "fasta_lower_case_with_n_runs": ["-f", fasta("l1.fa", src1, 2000, lower=True), "-f2", fasta("l2.fa", src2, 2000, lower=True)],
        "fasta_gz_pe": ["-f", fasta("g1.fa.gz", src1, 2000, gz=True), "-f2", fasta("g2.fa.gz", src2, 2000, gz=True)],
        "fasta_and_fastq": ["-f", fasta("m1.fa", src1, 2000), "-f2", a2],
    })
    for name, args in cases.items():
        outs = []
        for binary, extra in ((ref_bin, ["-t", "1"]), (host_oracle_binary, ["-t", "4"])):
            out = str(tmp_path / (name + "_" + os.path.basename(binary) + ".sam"))
            r = subprocess.run([binary, "-silent", "-i", SMALL_PREFIX] + args + ["-o", out] + extra, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
            outs.append((r.returncode, open(out, "rb").read()))
        assert outs[0] == outs[1], name


@pytest.mark.parametrize("threads", [1, 3, 16])
def test_output_does_not_depend_on_the_thread_count(threads, host_oracle_binary, tmp_path):
    for case in ("pe", "edge_pe", "pacbio"):
        got, want, _ = run_case(host_oracle_binary, case, str(tmp_path), ["-t", str(threads)])
        assert got == want, (case, threads)


def test_cli_error_behaviour(host_oracle_binary, tmp_path):
    """exit codes and messages of the command line for bad invocations -- the values below are what the unmodified
    reference binary prints/returns for the same arguments (src/main.cpp:106-207); only the banner line differs"""
    fq = materialise(str(tmp_path), "pe_1.fq")

    def run(*args):
        r = subprocess.run([host_oracle_binary] + list(args), stdout=subprocess.PIPE, stderr=subprocess.PIPE, cwd=str(tmp_path))
        return r.returncode, r.stdout.decode(), r.stderr.decode()

    rc, out, _ = run()
    assert rc == 0 and "Usage:" in out and "-i Index_Prefix" in out
    rc, out, _ = run("-v")
    assert rc == 0 and out == "kart v2.5.6\n\n"
    rc, out, _ = run("-i", "/nonexistent/idx", "-f", fq)
    assert rc == 1 and out.startswith("Error! Please specify a valid reference index!\n") and "Usage:" in out
    rc, out, _ = run("-i", SMALL_PREFIX)
    assert rc == 1 and "Usage:" in out
    rc, out, _ = run("-i", SMALL_PREFIX, "-f", "/nonexistent.fq", "-o", str(tmp_path / "x.sam"))
    assert rc == 0 and out == "Cannot access file:[/nonexistent.fq]\n"
    rc, out, _ = run("-i", SMALL_PREFIX, "-f", fq, "-f2", fq, "-f2", fq, "-o", str(tmp_path / "x.sam"))
    assert rc == 1 and out == "Error! Paired-end reads input numbers do not match!\nRead1:\n\t%s\nRead2:\n\t%s\n\t%s\n" % (fq, fq, fq)
    rc, out, err = run("-i", SMALL_PREFIX, "-f", fq, "-o", "/nonexistent_dir/x.sam")
    assert "Cannot open file [/nonexistent_dir/x.sam]" in err


def test_output_errors_are_a_status_not_an_exit_inside_the_library(host_oracle_binary, tmp_path):
    """a write error (ENOSPC from /dev/full) and an output in tmpfs that the host's memory cannot hold end the run with a message
    and a non-zero status through run_mapping()'s return value (the free-space check used to call exit(), and it
    tripped on file systems that report no size)"""
    args = [materialise(str(tmp_path), a) if a.endswith((".fq", ".fa", ".gz")) else a for a in CASES["pe"]]
    r = subprocess.run([host_oracle_binary, "-silent", "-t", "4", "-i", SMALL_PREFIX] + args + ["-o", "/dev/full"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode == 1 and b"Error! write to the output" in r.stderr, (r.returncode, r.stderr[-300:])
    if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK):
        out = "/dev/shm/kart_test_nomem_%d.sam" % os.getpid()
        try:
            env = dict(os.environ, KART_AMD_MEM_RESERVE_MB=str(1 << 40))       # nobody has that much: the gate must trip, politely
            r = subprocess.run([host_oracle_binary, "-silent", "-t", "4", "-i", SMALL_PREFIX] + args + ["-o", out], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300, env=env)
            assert r.returncode == 1 and b"the host cannot hold it" in r.stderr, (r.returncode, r.stderr[-300:])
            r = subprocess.run([host_oracle_binary, "-silent", "-t", "4", "-i", SMALL_PREFIX] + args + ["-o", out], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
            assert r.returncode == 0 and open(out, "rb").read() == gzip.open(os.path.join(SAM, "pe.sam.gz")).read()
        finally:
            if os.path.exists(out):
                os.remove(out)


@pytest.mark.parametrize("writers,pwriters", [(1, 1), (2, 2), (4, 1), (1, 0)])
def test_writer_thread_mix_gives_the_same_file(writers, pwriters, host_oracle_binary, tmp_path):
    """the output through any mix of mapping threads and pwrite() threads (round 4: seven + one by default from -t 16 on) is the
    same file -- chunks of 4000 reads land at their offsets whoever copies them"""
    for case in ("pe", "se_m"):
        got, want, _ = run_case_env(host_oracle_binary, case, str(tmp_path), ["-t", "8"], {"KART_AMD_WRITER_THREADS": str(writers), "KART_AMD_PWRITE_THREADS": str(pwriters)})
        assert got == want, (case, writers, pwriters)


def test_output_to_a_pipe(host_oracle_binary, tmp_path):
    """a FIFO cannot be seeked: the writer must fall back from parallel pwrite to one sequential stream"""
    import threading
    fifo = str(tmp_path / "out.fifo")
    os.mkfifo(fifo)
    got = []
    t = threading.Thread(target=lambda: got.append(open(fifo, "rb").read()))
    t.start()
    args = [materialise(str(tmp_path), a) if a.endswith((".fq", ".fa", ".gz")) else a for a in CASES["pe"]]
    r = subprocess.run([host_oracle_binary, "-silent", "-t", "4", "-i", SMALL_PREFIX] + args + ["-o", fifo], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    t.join(60)
    assert r.returncode == 0, r.stdout.decode()[-500:]
    assert got and got[0] == gzip.open(os.path.join(SAM, "pe.sam.gz")).read()


def test_multi_hit_flags_match_live_reference_where_assigned(host_oracle_binary, tmp_path):
    """-m paired-end: the reference prints an uninitialised SamFlag for some secondary records.
    The pipeline marks exactly those with KART_AMD_UNSET_FLAG; every other FLAG -- and every other column of every
    record -- must equal the live reference's."""
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "kart")
    if not os.path.exists(ref_bin):
        pytest.skip("oracle/_ref/kart not present on this machine")
    from kart_amd import synth
    from kart_amd.index_build import read_fasta
    genome = {n: s for n, _, s in read_fasta(os.path.join(GOLDEN, "small.fa"))}
    names, r1, r2 = synth.simulate_pairs(genome, 12000, seed=78, err=0.02, mut=0.003, indel_frac=0.3, n_frac=0.0005)
    f1, f2 = str(tmp_path / "a_1.fq"), str(tmp_path / "a_2.fq")
    synth.write_fastq(f1, names, r1, mate=1)
    synth.write_fastq(f2, names, r2, mate=2)
    ref_lines, never = reference_sam_and_never_assigned_flags(ref_bin, ["-i", SMALL_PREFIX, "-f", f1, "-f2", f2, "-m"], str(tmp_path))
    out = str(tmp_path / "host.sam")
    subprocess.run([host_oracle_binary, "-silent", "-i", SMALL_PREFIX, "-f", f1, "-f2", f2, "-m", "-o", out, "-t", "4"], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=dict(os.environ, KART_AMD_UNSET_FLAG=str(UNSET_FLAG)))
    assert_sam_equals_reference_with_its_own_mask(ref_lines, never, open(out, "rb").read())

"""GPU: the product binary kart_amd/bin/kart-amd (host pipeline + HIP kernels through the C ABI) must
write SAM byte-identical to the reference: against the committed golden files, and -- where the
unmodified reference binary travelled with the snapshot (oracle/_ref/kart) -- against a live
`kart -t 1` run on a larger seeded input."""
import gzip
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, SMALL_PREFIX
from test_host_pipeline import (CASES, UNSET_FLAG, assert_sam_equals_reference_with_its_own_mask, assert_same_sam_up_to_unset_flags,
                                reference_sam_and_never_assigned_flags, run_case, run_case_env)

pytestmark = pytest.mark.gpu
KART_AMD = os.path.join(ROOT, "kart_amd", "bin", "kart-amd")
KART_REF = os.path.join(ROOT, "oracle", "_ref", "kart")


@pytest.fixture(scope="module")
def product_binary(built_lib):
    assert os.path.exists(KART_AMD), "kart_amd/bin/kart-amd missing: __graft_entry__.build() builds it"
    return KART_AMD


@pytest.mark.parametrize("case", sorted(CASES))
def test_golden_sam(case, product_binary, tmp_path):
    got, want, _ = run_case(product_binary, case, str(tmp_path))
    assert got == want


@pytest.mark.parametrize("env", [{"KG_DBG_JOB_CAPACITY": "9"}, {"KG_DBG_OPS_CAPACITY": "150"}, {"KG_DBG_SPILL_CAPACITY": "5"},
                                 {"KG_DBG_JOB_CAPACITY": "40", "KG_DBG_OPS_CAPACITY": "700", "KG_DBG_SPILL_CAPACITY": "30"}])
def test_full_lists_hand_their_reads_to_the_host(env, product_binary, tmp_path):
    """aln_plan_kernel / aln_partition_kernel with the spill, job or op-byte list full (lists as short as a few entries, abi.hip
    KG_DBG_*_CAPACITY): the candidates beyond go back to the host, and what they had already taken INSIDE the lists is left as an
    empty entry -- aln_finish_kernel and the NW kernels walk every entry below the counters.  Several batches on one workspace, so
    that a stale entry of the batch before would be found.  Same SAM as the reference's."""
    for case in ("pe", "pe_m"):
        if case not in CASES:
            continue
        got, want, log = run_case_env(product_binary, case, str(tmp_path), ["-t", "1"], dict(env, KART_AMD_BATCH_READS="4000"))
        assert got == want, (case, env)


@pytest.mark.parametrize("flags", [[], ["-m"], ["-g", "40"]])       # (-g 40: MaxGaps beyond what the packed partition scan takes -- its scalar form)
def test_live_reference_30k_pairs(flags, product_binary, tmp_path):
    assert os.path.exists(KART_REF), "oracle/_ref/kart did not travel to the GPU box: __graft_entry__.build() makes it where /root/reference exists"
    from kart_amd import synth
    from kart_amd.index_build import read_fasta
    genome = {n: s for n, _, s in read_fasta(os.path.join(GOLDEN, "small.fa"))}
    names, r1, r2 = synth.simulate_pairs(genome, 30000, seed=77, err=0.02, mut=0.003, indel_frac=0.3, n_frac=0.0005)
    f1, f2 = str(tmp_path / "a_1.fq"), str(tmp_path / "a_2.fq")
    synth.write_fastq(f1, names, r1, mate=1)
    synth.write_fastq(f2, names, r2, mate=2)
    # the records whose FLAG the reference never assigns come from the reference alone (two runs under different heap fill bytes);
    # the product prints UNSET_FLAG on exactly those
    ref_lines, never = reference_sam_and_never_assigned_flags(KART_REF, ["-i", SMALL_PREFIX, "-f", f1, "-f2", f2] + flags, str(tmp_path))
    out = str(tmp_path / "amd.sam")
    subprocess.run([product_binary, "-silent", "-i", SMALL_PREFIX, "-f", f1, "-f2", f2, "-o", out] + flags, check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=dict(os.environ, KART_AMD_UNSET_FLAG=str(UNSET_FLAG)))
    got = open(out, "rb").read()
    masked = assert_sam_equals_reference_with_its_own_mask(ref_lines, never, got)
    if flags != ["-m"]:
        assert masked == 0          # without -m every printed FLAG is assigned by the reference: byte identity
        assert got == b"\n".join(ref_lines)


def test_live_reference_pacbio_7kb(product_binary, tmp_path):
    """configs[3] in miniature: 7 kb reads at 15 % error with -pacbio (SensitiveMode seeding, recursive 8-mer
    partition, NW fragments of several hundred bases -> the wave-per-pair kernel with multiple stripes)."""
    assert os.path.exists(KART_REF), "oracle/_ref/kart did not travel to the GPU box: __graft_entry__.build() makes it where /root/reference exists"
    from kart_amd import synth
    from kart_amd.index_build import read_fasta
    genome = {n: s for n, _, s in read_fasta(os.path.join(GOLDEN, "small.fa"))}
    names, reads = synth.simulate_long_reads(genome, 300, seed=31, read_len=7000, err=0.15, indel_err_frac=0.1)
    fq = str(tmp_path / "long.fq")
    synth.write_fastq(fq, names, reads)
    outs = []
    for binary in (KART_REF, product_binary):
        out = str(tmp_path / (os.path.basename(binary) + ".sam"))
        extra = ["-t", "1"] if binary == KART_REF else ["-t", "8"]
        subprocess.run([binary, "-silent", "-i", SMALL_PREFIX, "-f", fq, "-pacbio", "-o", out] + extra, check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        outs.append(open(out, "rb").read())
    assert outs[0] == outs[1]
    # the long-read pipeline under small batches and slices (several batches, several fragment calls per batch, 1 .. 3 of them in
    # flight, batches doubling), with the fragment pairs planned by the host, and with IdentifyNormalPairs on lane 0 only: same bytes
    for env in ({"KART_AMD_PACBIO_CHUNKS": "8", "KART_AMD_FRAG_SLICE": "4", "KART_AMD_FRAG_DEPTH": "1"},
                {"KART_AMD_PACBIO_CHUNKS": "8", "KART_AMD_FRAG_SLICE": "4", "KART_AMD_FRAG_DEPTH": "2"},
                {"KART_AMD_PACBIO_CHUNKS": "4", "KART_AMD_PACBIO_MAX_CHUNKS": "16", "KART_AMD_FRAG_SLICE": "6", "KART_AMD_FRAG_DEPTH": "3"},
                {"KART_AMD_HOST_FRAGMENTS": "1"}, {"KG_FRAG_NO_FAST_PAIRS": "1"}):
        out = str(tmp_path / "variant.sam")
        r = subprocess.run([product_binary, "-silent", "-i", SMALL_PREFIX, "-f", fq, "-pacbio", "-o", out, "-t", "2"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                           env=dict(os.environ, **env))
        assert r.returncode == 0, (env, r.stdout.decode()[-400:])
        assert open(out, "rb").read() == outs[0], env


def test_live_reference_pacbio_unseeded_stretch_over_7000(product_binary, tmp_path):
    """a long read whose middle 7600 bases are N: no seeds and no 8-mers there, so the whole stretch reaches nw_alignment as one
    7600 x ~7600 fragment -- beyond what the wave-per-pair kernel's LDS holds (the run used to exit with "kg_nw_batch")"""
    assert os.path.exists(KART_REF), "oracle/_ref/kart did not travel to the GPU box: __graft_entry__.build() makes it where /root/reference exists"
    from kart_amd.index_build import read_fasta
    genome = {n: s for n, _, s in read_fasta(os.path.join(GOLDEN, "small.fa"))}
    chrom = max(genome.values(), key=len)
    chrom = bytes(chrom)
    assert len(chrom) > 40000
    reads = []
    for x, half, gap in ((2500, 800, 7600), (4000, 1500, 7100), (8500, 400, 8200)):    # (chosen so that the seeds of both halves chain into one candidate)
        reads.append(chrom[x:x + half] + b"N" * gap + chrom[x + half + gap:x + 2 * half + gap])
    reads.append(chrom[30000:33000])
    fq = str(tmp_path / "nrun.fq")
    with open(fq, "wb") as fh:
        for i, r in enumerate(reads):
            fh.write(b"@n%d\n" % i + r + b"\n+\n" + b"5" * len(r) + b"\n")
    outs = []
    for binary in (KART_REF, product_binary):
        out = str(tmp_path / (os.path.basename(binary) + ".sam"))
        extra = ["-t", "1"] if binary == KART_REF else ["-t", "4"]
        r = subprocess.run([binary, "-silent", "-i", SMALL_PREFIX, "-f", fq, "-pacbio", "-o", out] + extra, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        assert r.returncode == 0, r.stdout.decode()[-400:]
        outs.append(open(out, "rb").read())
    assert outs[0] == outs[1]
    cigars = [l.split(b"\t")[5] for l in outs[0].splitlines() if not l.startswith(b"@")]
    assert b"7599I7599D" in cigars[0] and b"7099I7099D" in cigars[1] and b"8199I8199D" in cigars[2], cigars   # the stretch went through nw_alignment, not a clip


def test_long_read_report_is_the_devices(product_binary, tmp_path):
    """kg_longread_batch (src/Mapping.cpp:513-530 on the device: IdentifyNormalPairs of the read, pass 1, the fragment kernels on
    device-resident requests, pass 2 off the op strings, CIGAR pool, flag, MAPQ): 400 x 7 kb reads at 15 % error plus reads that must
    go back to the host (a literal '-', which the reference's scans take for a gap column) and reads with N runs -- identical to the
    live reference, nearly every read decided on the device, every device record equal to the host's text (KART_AMD_CHECK_ALIGN),
    the same bytes with the report forced onto the host, under small batches, and with mate files (printed by the paired function:
    the host's report)"""
    assert os.path.exists(KART_REF), "oracle/_ref/kart did not travel to the GPU box: __graft_entry__.build() makes it where /root/reference exists"
    from kart_amd import synth
    from kart_amd.index_build import read_fasta
    genome = {n: s for n, _, s in read_fasta(os.path.join(GOLDEN, "small.fa"))}
    names, reads = synth.simulate_long_reads(genome, 400, seed=77, read_len=7000, err=0.15, indel_err_frac=0.1)
    names2, reads2 = synth.simulate_long_reads(genome, 60, seed=78, read_len=3000, err=0.02, indel_err_frac=0.3)      # accurate reads: long exact matches, the <= 2-mismatch shortcut
    names, reads = list(names) + ["acc_" + n for n in names2], [np.array(r, dtype=np.uint8) for r in reads] + [np.array(r, dtype=np.uint8) for r in reads2]
    for i in (3, 50, 120):                                  # literal dashes -> KG_ALN_HOST
        reads[i][1000 + i] = ord("-"); reads[i][2000] = ord("-")
    for i in (7, 90):                                       # N runs inside the read (fragment kernels hand such fragments back)
        reads[i][3000:3040] = ord("N")
    reads[200][10] = ord("n"); reads[200][4000] = ord("a") if reads[200][4000] != ord("A") else ord("c")     # lower case: raw-character comparisons
    fq = str(tmp_path / "long.fq")
    synth.write_fastq(fq, names, reads)
    ref_out = str(tmp_path / "ref.sam")
    subprocess.run([KART_REF, "-silent", "-i", SMALL_PREFIX, "-f", fq, "-pacbio", "-o", ref_out, "-t", "1"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    want = open(ref_out, "rb").read()
    out = str(tmp_path / "amd.sam")
    log, err = _run_verbose(product_binary, ["-f", fq, "-pacbio"], out, {})
    assert open(out, "rb").read() == want
    m = re.search(r"device report: (\d+) reads decided on the device, (\d+) mapped by the host stages", log)
    assert m and int(m.group(1)) + int(m.group(2)) == len(reads), log[-600:]
    assert int(m.group(1)) >= 0.9 * len(reads) and int(m.group(2)) >= 3, log[-600:]
    log, err = _run_verbose(product_binary, ["-f", fq, "-pacbio"], out, {"KART_AMD_CHECK_ALIGN": "1"})
    line = [l for l in log.splitlines() if l.startswith("CHECK_ALIGN")]
    assert line and line[0].endswith(" 0 differ") and not line[0].startswith("CHECK_ALIGN: 0 device"), (line, err[:800])
    assert open(out, "rb").read() == want
    # (the last one: ONE workspace for seeding and report, many small batches -- the switch used to seed the next batch into the arrays the
    #  report was still reading, and the run hung; profiles/r05za_jobs.log)
    for env in ({"KART_AMD_HOST_LONG": "1"}, {"KART_AMD_PACBIO_CHUNKS": "4", "KART_AMD_FRAG_SLICE": "4"}, {"KART_AMD_PACBIO_CHUNKS": "8", "KART_AMD_PACBIO_MAX_CHUNKS": "32", "KG_FRAG_NO_FAST_PAIRS": "1"},
                {"KART_AMD_LONG_NO_OVERLAP": "1", "KART_AMD_PACBIO_CHUNKS": "4", "KART_AMD_PACBIO_MAX_CHUNKS": "8"}):
        log, err = _run_verbose(product_binary, ["-f", fq, "-pacbio"], out, env)
        assert open(out, "rb").read() == want, env
        if "KART_AMD_HOST_LONG" in env:
            assert "device report: 0 reads decided on the device" in log


@pytest.mark.parametrize("threads", [1, 3, 16, 64])
def test_output_does_not_depend_on_the_thread_count(threads, product_binary, tmp_path):
    for case in ("pe", "pe_m", "pacbio"):
        got, want, _ = run_case(product_binary, case, str(tmp_path), ["-t", str(threads)])
        assert got == want, (case, threads)


def test_repeated_runs_are_deterministic(product_binary, tmp_path):
    """Regression test for an intermittent race (NW work lists emptied by a late memset of recycled
    stream-ordered pool memory): the same paired-end input mapped 12 times must give the golden SAM every time."""
    for it in range(12):
        got, want, _ = run_case(product_binary, "pe_g2" if it % 2 else "pe", str(tmp_path))
        assert got == want, "run %d differs" % it


def test_bam_output_of_the_product_binary(product_binary, tmp_path):
    """-bo through the HIP-backed binary: decodes to the golden SAM records"""
    from test_bam_output import decode_bam, sam_records
    from test_host_pipeline import materialise
    args = [materialise(str(tmp_path), a) if a.endswith((".fq", ".fa", ".gz")) else a for a in CASES["pe"]]
    bam, sam = str(tmp_path / "o.bam"), str(tmp_path / "gold.sam")
    r = subprocess.run([product_binary, "-silent", "-t", "8", "-i", SMALL_PREFIX] + args + ["-bo", bam], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()[-400:]
    open(sam, "wb").write(gzip.open(os.path.join(GOLDEN, "sam", "pe.sam.gz")).read())
    head, want = sam_records(sam)
    text, _, got = decode_bam(bam)
    assert text == head and got == want


def _device_lists():
    """device lists for the sharded runs: several processes on device 0 always (a 1-GPU box exercises the whole mechanism
    that way), plus every real multi-device split the box offers"""
    from kart_amd import api
    n = api.device_count()
    lists = ["0,0", "0,0,0"]
    for k in (2, 4, 8):
        if n >= k:
            lists.append(",".join(str(i) for i in range(k)))
    return lists


@pytest.mark.parametrize("case", ["pe_plain", "pe_interleaved", "edge_se", "pe"])
def test_sharded_golden_sam(case, product_binary, tmp_path):
    """one process per listed device, contiguous chunk ranges of the library (detail/shard.inc): byte-identical SAM"""
    for devices in _device_lists():
        got, want, _ = run_case(product_binary, case, str(tmp_path), ["-gpu", devices, "-t", "8"])
        assert got == want, (case, devices)


def test_sharded_live_reference_160k_reads(product_binary, tmp_path):
    """40 chunks with drifting insert sizes (the estimate keeps moving): 1, 2, 3 processes -- and 2/4/8 real devices where the
    box has them -- all equal to the reference's -t 1"""
    assert os.path.exists(KART_REF), "oracle/_ref/kart did not travel to the GPU box: __graft_entry__.build() makes it where /root/reference exists"
    from kart_amd import synth
    from kart_amd.index_build import read_fasta
    genome = {n: s for n, _, s in read_fasta(os.path.join(GOLDEN, "small.fa"))}
    f1, f2 = str(tmp_path / "s_1.fq"), str(tmp_path / "s_2.fq")
    with open(f1, "wb") as o1, open(f2, "wb") as o2:
        for part, ins in enumerate((300, 260, 220, 180)):
            names, r1, r2 = synth.simulate_pairs(genome, 20000, seed=950 + part, err=0.02, mut=0.002, indel_frac=0.3, ins_mean=float(ins), ins_sd=ins / 8.0)
            names = ["p%d_%s" % (part, n) for n in names]
            p1, p2 = str(tmp_path / "t1.fq"), str(tmp_path / "t2.fq")
            synth.write_fastq(p1, names, r1, mate=1)
            synth.write_fastq(p2, names, r2, mate=2)
            o1.write(open(p1, "rb").read())
            o2.write(open(p2, "rb").read())
    ref = str(tmp_path / "ref.sam")
    subprocess.run([KART_REF, "-silent", "-i", SMALL_PREFIX, "-f", f1, "-f2", f2, "-o", ref, "-t", "1"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    want = open(ref, "rb").read()
    for devices in ["0"] + _device_lists():
        out = str(tmp_path / "o.sam")
        r = subprocess.run([product_binary, "-silent", "-i", SMALL_PREFIX, "-f", f1, "-f2", f2, "-o", out, "-gpu", devices, "-t", "8"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        assert r.returncode == 0, (devices, r.stdout.decode()[-600:])
        assert open(out, "rb").read() == want, devices


def _run_verbose(binary, args, out, env=None):
    r = subprocess.run([binary, "-silent", "-i", SMALL_PREFIX] + args + ["-o", out, "-t", "8"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=dict(os.environ, KART_AMD_VERBOSE="1", **(env or {})), timeout=900)          # (a hang is a failure, not a stalled suite)
    assert r.returncode == 0, r.stderr.decode()[-600:]
    return r.stdout.decode(), r.stderr.decode()


@pytest.mark.parametrize("case", ["pe", "pe_g2", "se", "edge_pe", "edge_se", "pe_m", "se_m", "edge_se_m", "pacbio"])
def test_device_report_equals_the_host_report_record_by_record(case, product_binary, tmp_path):
    """kg_align_batch (pairing, mate rescue, normal pairs, 8-mer partition, NW, CIGAR, flags, MAPQ on the device) against the host
    implementation of the same reference code: KART_AMD_CHECK_ALIGN maps every read on the host as well and compares the SAM text
    made from each device record with the host's; and with the report forced onto the host (KART_AMD_HOST_ALIGN) the golden SAM
    still comes out"""
    from test_host_pipeline import materialise
    args = [materialise(str(tmp_path), a) if a.endswith((".fq", ".fa", ".gz")) else a for a in CASES[case]]
    out = str(tmp_path / "o.sam")
    log, err = _run_verbose(product_binary, args, out, {"KART_AMD_CHECK_ALIGN": "1"})
    line = [l for l in log.splitlines() if l.startswith("CHECK_ALIGN")]
    assert line and line[0].endswith(" 0 differ") and not line[0].startswith("CHECK_ALIGN: 0 device"), (line, err[:800])
    want = gzip.open(os.path.join(GOLDEN, "sam", case + ".sam.gz")).read()
    assert open(out, "rb").read() == want
    log, _ = _run_verbose(product_binary, args, out, {"KART_AMD_HOST_ALIGN": "1"})
    assert open(out, "rb").read() == want
    assert "device report: 0 reads decided on the device" in log


@pytest.mark.parametrize("sa", ["compact", "dense4", "dense8", "sampled"])
@pytest.mark.parametrize("case", ["pe", "edge_pe", "se_m", "pacbio"])
def test_smaller_index_modes_give_the_same_sam(case, sa, product_binary, tmp_path):
    """KART_AMD_SA: the compact index (under KG_FORCE_U64 its 5-byte entries), every 4th / 8th suffix-array entry resident (searches walk to a sampled row, then finish against the text), or only
    the file's samples (no text finishing): the golden SAM byte for byte, and the load reports the mode"""
    from test_host_pipeline import materialise
    args = [materialise(str(tmp_path), a) if a.endswith((".fq", ".fa", ".gz")) else a for a in CASES[case]]
    out = str(tmp_path / "o.sam")
    want = gzip.open(os.path.join(GOLDEN, "sam", case + ".sam.gz")).read()
    for env in ({}, {"KG_FORCE_U64": "1"}):
        _run_verbose(product_binary, args, out, dict(env, KART_AMD_SA=sa))
        assert open(out, "rb").read() == want, env
    r = subprocess.run([product_binary, "-silent", "-i", SMALL_PREFIX] + args + ["-o", out], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, KART_AMD_SA="half"))
    assert r.returncode != 0 and b"KART_AMD_SA" in r.stdout + r.stderr


def test_device_report_on_30k_live_pairs_with_rescue_and_indels(product_binary, tmp_path):
    """30 k pairs at 2 % error with indels and short inserts (rescue windows, gap fragments of every size, estimate below
    MaxInsertSize): most reads are decided on the device, every device record equals the host's text, and the SAM equals the
    live reference's"""
    from kart_amd import synth
    from kart_amd.index_build import read_fasta
    genome = {n: s for n, _, s in read_fasta(os.path.join(GOLDEN, "small.fa"))}
    names, r1, r2 = synth.simulate_pairs(genome, 30000, seed=177, err=0.02, mut=0.003, indel_frac=0.3, n_frac=0.0005, ins_mean=400.0, ins_sd=50.0)
    f1, f2 = str(tmp_path / "a_1.fq"), str(tmp_path / "a_2.fq")
    synth.write_fastq(f1, names, r1, mate=1)
    synth.write_fastq(f2, names, r2, mate=2)
    out = str(tmp_path / "o.sam")
    log, err = _run_verbose(product_binary, ["-f", f1, "-f2", f2], out, {"KART_AMD_CHECK_ALIGN": "1"})
    line = [l for l in log.splitlines() if l.startswith("CHECK_ALIGN")]
    assert line and line[0].endswith(" 0 differ"), (line, err[:800])
    log, _ = _run_verbose(product_binary, ["-f", f1, "-f2", f2], out)
    dev = [l for l in log.splitlines() if l.startswith("device report:") and "decided on the device" in l][0]
    n_dev = int(dev.split()[2])
    assert n_dev > 30000, dev                      # more than half of the 60 k reads
    if os.path.exists(KART_REF):
        ref = str(tmp_path / "ref.sam")
        subprocess.run([KART_REF, "-silent", "-i", SMALL_PREFIX, "-f", f1, "-f2", f2, "-o", ref, "-t", "1"], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        assert open(out, "rb").read() == open(ref, "rb").read()


def test_single_contig_genome_live_reference(product_binary, tmp_path):
    """a genome of ONE contig takes the n_chr == 1 branches of GenCoordinateInfo (src/AlignmentCandidates.cpp:523,541) -- on the
    device and on the host; reads keep 3 kb away from both ends (the reference's rescue windows index outside the text there, App. B-3)"""
    assert os.path.exists(KART_REF), "oracle/_ref/kart did not travel to the GPU box: __graft_entry__.build() makes it where /root/reference exists"
    from kart_amd import index_build, synth
    from kart_amd.index_build import read_fasta
    genome = {n: s for n, _, s in read_fasta(os.path.join(GOLDEN, "small.fa"))}
    chrom = max(genome.values(), key=len)
    fa = str(tmp_path / "one.fa")
    synth.write_fasta(fa, {"solo": chrom})
    prefix = str(tmp_path / "one")
    index_build.build_index(fa, prefix, device="cpu")
    inner = {"solo": chrom[3000:-3000]}
    names, r1, r2 = synth.simulate_pairs(inner, 9000, seed=5, err=0.02, mut=0.003, indel_frac=0.3, skip=())
    # the simulated fragment lies inside the trimmed copy: it is a substring of the full contig, so the reads map 3 kb further in
    f1, f2 = str(tmp_path / "a_1.fq"), str(tmp_path / "a_2.fq")
    synth.write_fastq(f1, names, r1, mate=1)
    synth.write_fastq(f2, names, r2, mate=2)
    outs = []
    for binary, extra in ((KART_REF, ["-t", "1"]), (product_binary, ["-t", "8"])):
        out = str(tmp_path / (os.path.basename(binary) + ".sam"))
        r = subprocess.run([binary, "-silent", "-i", prefix, "-f", f1, "-f2", f2, "-o", out] + extra, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                           env=dict(os.environ, KART_AMD_VERBOSE="1"))
        assert r.returncode == 0, r.stdout.decode()[-400:]
        outs.append(open(out, "rb").read())
        log = r.stdout.decode()
    assert outs[0] == outs[1]
    dev = [l for l in log.splitlines() if l.startswith("device report:") and "decided on the device" in l][0]
    assert int(dev.split()[2]) > 9000, dev




@pytest.mark.parametrize("case", ["pe", "pacbio", "edge_pe"])
def test_outgrown_workspaces_are_retired_not_freed(case, product_binary, tmp_path):
    """ADVICE round 2 (use-after-free): the backend's workspace starts far too small (KART_AMD_TINY_WORKSPACE), is outgrown while
    the pipeline ramps its batches up and while the page-locked results of earlier batches are still being read; those must
    survive (the outgrown workspace is retired, freed four batches later).  The host's reader / printer path (no device stream)."""
    from test_host_pipeline import materialise
    args = [materialise(str(tmp_path), a) if a.endswith((".fq", ".fa", ".gz")) else a for a in CASES[case]]
    out = str(tmp_path / "o.sam")
    for it in range(3):
        r = subprocess.run([product_binary, "-silent", "-i", SMALL_PREFIX, "-t", "6", "-o", out] + args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                           env=dict(os.environ, KART_AMD_TINY_WORKSPACE="1", KART_AMD_NO_STREAM="1", KART_AMD_BATCH_READS="4000"))
        assert r.returncode == 0, r.stdout.decode()[-600:]
        assert open(out, "rb").read() == gzip.open(os.path.join(GOLDEN, "sam", case + ".sam.gz")).read()


def test_sharded_parts_on_the_device(product_binary, tmp_path):
    """-gpu 0,0,0 -parts: three processes through the device stream, one output file each; concatenated = the reference's SAM"""
    from test_host_pipeline import materialise
    args = [materialise(str(tmp_path), a) if a.endswith((".fq", ".fa", ".gz")) else a for a in CASES["pe_plain"]]
    out = str(tmp_path / "o.sam")
    r = subprocess.run([product_binary, "-silent", "-i", SMALL_PREFIX, "-t", "6", "-o", out, "-gpu", "0,0,0", "-parts"] + args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()[-600:]
    got = b"".join(open("%s.%d" % (out, q), "rb").read() for q in range(3))
    assert got == gzip.open(os.path.join(GOLDEN, "sam", "pe.sam.gz")).read()

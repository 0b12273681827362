"""GPU: SAM identity with the live reference at the size BASELINE.json's metric is quoted on -- the hg38-SIZED synthetic index
(3.1 Gbp in 24 contigs behind the decoy, 45 % repeat families; bench.py's workload, same seed, same files).

One module-scoped fixture builds (or finds) the index, loads it in the compact mode (66.9 GB; same seeds and SAM as every other
mode, tests/test_sam_gpu.py::test_smaller_index_modes_give_the_same_sam), writes three read sets with bench.py's own generators
and starts the unmodified reference binary (oracle/_ref/kart -t 1) on all of them at once, each run one core:

  configs[2]  0.5 M reads, 150 bp paired-end, 1.1 % error            -> byte identity
  configs[4]  0.2 M reads at 2.1 % error with -m                      -> identity up to the FLAGs the reference never assigns; that
                                                                         set comes from the reference alone (two MALLOC_PERTURB_ runs)
                                                                         and the product's sentinel set must EQUAL it (App. B-12)
  configs[3]  20 000 x 7 kb at 15 % error with -pacbio                -> byte identity.  Single-end long reads are independent of
                                                                         each other, so the reference runs as EIGHT processes on disjoint
                                                                         slices of 2500 reads (1000 of 2 M reads was thin);
                                                                         a few reads carry N runs, lower-case letters and literal '-'
                                                                         (those go back to the host: KG_ALN_HOST)

Round 3 held these comparisons only inside bench.py, and lost them with the bench.  tmpfs use: index 5.4 GB + reads and SAM < 1 GB."""
import argparse
import os
import subprocess
import time

import pytest

from conftest import ROOT
from test_host_pipeline import UNSET_FLAG, assert_sam_equals_reference_with_its_own_mask

pytestmark = pytest.mark.gpu
KART_REF = os.path.join(ROOT, "oracle", "_ref", "kart")
N_LONG, LONG_SLICES = int(os.environ.get("KART_TEST_N_LONG", "20000")), int(os.environ.get("KART_TEST_LONG_SLICES", "8"))      # (a longer run by hand: 100 000 reads in 16 slices)


def odd_long_reads(path, n, read_len):
    """a few records of bench.write_long_reads' file get what real long reads have and the generator has next to none of: N runs (no seeds, no
    8-mers across them: the fragment kernels hand such fragments back), lower-case letters (raw-character comparisons) and a literal '-'
    (the reference's CIGAR scans take it for a gap column: such a read is the host's, KG_ALN_HOST)"""
    import numpy as np
    from benchkit.reads import record_starts
    seq_start, seq_len = record_starts(path)
    assert len(seq_start) == n and (seq_len == read_len).all()
    mm = np.memmap(path, dtype=np.uint8, mode="r+")
    rng = np.random.default_rng(5)
    for i in rng.choice(n, 60, replace=False):
        at = int(seq_start[int(i)])
        kind = int(i) % 3
        if kind == 0:
            p = int(rng.integers(100, read_len - 200))
            mm[at + p:at + p + int(rng.integers(1, 60))] = ord("N")
        elif kind == 1:
            for p in rng.integers(0, read_len, 5):
                mm[at + int(p)] = mm[at + int(p)] | 0x20
        else:
            mm[at + int(rng.integers(0, read_len))] = ord("-")
    mm.flush()
    del mm


@pytest.fixture(scope="module")
def hg38(built_lib):
    import torch
    import bench
    from kart_amd import api
    assert os.path.exists(KART_REF), "oracle/_ref/kart did not travel to the GPU box: __graft_entry__.build() makes it where /root/reference exists"
    mem = bench.host_memory()
    # a box too small for this module is a RED test, not a silent skip: the hg38-size parity is the headline's
    # parity; KART_ALLOW_SMALL_BOX=1 is the explicit way to run the rest of the suite on a smaller lease
    small = []
    if (mem.get("usable") or 0) < (170 << 30):
        small.append("the hg38-sized comparison needs ~130 GB of host memory (index files, eleven reference processes of ~11 GB); usable: %s" % mem.get("usable"))
    if torch.cuda.get_device_properties(0).total_memory < (100 << 30):
        small.append("the hg38-sized index needs 67 GB of device memory")
    if small:
        if os.environ.get("KART_ALLOW_SMALL_BOX") == "1":
            pytest.skip("; ".join(small) + " (KART_ALLOW_SMALL_BOX=1)")
        pytest.fail("; ".join(small) + " -- set KART_ALLOW_SMALL_BOX=1 to skip the hg38-size parity on purpose")
    dev = torch.device("cuda", 0)
    assert api.device_count() > 0
    args = argparse.Namespace(genome_len=bench.HG38_LEN, bucketed=None, repeat_frac=0.45)
    workdir = bench.pick_workdir(12 << 30)
    os.makedirs(workdir, exist_ok=True)
    t0 = time.time()
    prefix, codes, _ = bench.prepare_index(args, dev, 0, workdir, lambda: None)
    t_index = time.time() - t0
    tag = os.path.join(workdir, "t38_%d" % os.getpid())
    files = {"pe": (tag + "_pe_1.fq", tag + "_pe_2.fq"), "mh": (tag + "_mh_1.fq", tag + "_mh_2.fq"), "long": (tag + "_long.fq",)}
    # bench.py's generators (wgsim's model incl. haplotype indels, benchkit/reads.py) with the seeds and rates of its timed files, configs[4] and configs[3]
    gen = {"pe": bench.write_fastq_pairs(codes, 250_000, 5, files["pe"][0], files["pe"][1], dev, err=0.01),
           "mh": bench.write_fastq_pairs(codes, 100_000, 41, files["mh"][0], files["mh"][1], dev, err=0.02),
           "long": bench.write_long_reads(codes, N_LONG, 7000, 31, files["long"][0], dev)}
    bench.release_haplotypes()
    assert gen["pe"]["pairs_with_indel"] > 0.02 * 250_000 and gen["long"]["reads_with_indel"] > 0.4 * N_LONG, gen      # the reads DO carry wgsim's indels
    odd_long_reads(files["long"][0], N_LONG, 7000)
    slices = [tag + "_long_%d.fq" % k for k in range(LONG_SLICES)]
    bench.split_records(files["long"][0], slices, N_LONG // LONG_SLICES)
    del codes
    torch.cuda.empty_cache()
    assert api.device_count() > 0, "the library lost the device after the read generators"

    # the reference, four runs at once (-t 1 each: with more threads it prints the chunks in completion order)
    def ref(name, flags, inputs, env=None):
        out = tag + "_ref_%s.sam" % name
        a = [KART_REF, "-silent", "-t", "1", "-i", prefix, "-f", inputs[0]] + (["-f2", inputs[1]] if len(inputs) > 1 else []) + flags + ["-o", out]
        return out, subprocess.Popen(a, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=dict(os.environ, **(env or {})))

    perturb = lambda b: {"MALLOC_PERTURB_": str(b), "GLIBC_TUNABLES": "glibc.malloc.tcache_count=0"}
    t0 = time.time()
    procs = {"pe": ref("pe", [], files["pe"]), "mh85": ref("mh85", ["-m"], files["mh"], perturb(85)), "mh170": ref("mh170", ["-m"], files["mh"], perturb(170)),
             }
    for k, sl in enumerate(slices):
        procs["long_%d" % k] = ref("long_%d" % k, ["-pacbio"], (sl,))

    assert api.device_count() > 0, "the library lost the device after starting the reference processes"
    # the product meanwhile: one session, the three runs through the host library (kh_map)
    os.environ["KART_AMD_SA"] = "compact"
    try:
        sess = api.HostSession(prefix, 0, 8)
    finally:
        del os.environ["KART_AMD_SA"]
    got, stats = {}, {}
    for name, flags, inputs, env in (("pe", [], files["pe"], {}), ("mh", ["-m"], files["mh"], {"KART_AMD_UNSET_FLAG": str(UNSET_FLAG)}), ("long", ["-pacbio"], files["long"], {})):
        out = tag + "_amd_%s.sam" % name
        os.environ.update(env)
        try:
            stats[name] = sess.map(["-silent", "-f", inputs[0]] + (["-f2", inputs[1]] if len(inputs) > 1 else []) + flags + ["-o", out])
        finally:
            for k in env:
                del os.environ[k]
        got[name] = open(out, "rb").read()
        os.remove(out)
    sess.close()
    want = {}
    for name, (out, p) in procs.items():
        rc = p.wait(timeout=1500)
        assert rc == 0, "the reference failed on %s (status %d)" % (name, rc)
        want[name] = open(out, "rb").read()
        os.remove(out)
    # the slices' records in order behind one header = what one process prints for the whole file
    parts = [want.pop("long_%d" % k) for k in range(LONG_SLICES)]
    header = b"".join(l for l in parts[0].splitlines(True) if l.startswith(b"@"))
    want["long"] = header + b"".join(b"".join(l for l in p_.splitlines(True) if not l.startswith(b"@")) for p_ in parts)
    for sl in slices:
        os.remove(sl)
    for fs in files.values():
        for f in fs:
            os.remove(f)
    print("hg38-sized parity: index %.0f s, reference runs %.0f s" % (t_index, time.time() - t0))
    return {"got": got, "want": want, "stats": stats}


def test_configs2_default_500k_reads_identical(hg38):
    assert hg38["stats"]["pe"].total_reads == 500_000 and hg38["stats"]["pe"].stream_reads > 0          # (through the device stream: FASTQ text in, SAM text out)
    assert hg38["got"]["pe"] == hg38["want"]["pe"]


def test_configs4_multi_hit_200k_reads_identical_up_to_never_assigned_flags(hg38):
    a, b = hg38["want"]["mh85"].split(b"\n"), hg38["want"]["mh170"].split(b"\n")
    assert len(a) == len(b)
    never = set()
    for i, (x, y) in enumerate(zip(a, b)):
        if x != y:
            fx, fy = x.split(b"\t"), y.split(b"\t")
            assert fx[:1] + fx[2:] == fy[:1] + fy[2:], (x[:120], y[:120])      # only the FLAG may depend on the heap
            never.add(i)
    assert hg38["stats"]["mh"].total_reads == 200_000
    masked = assert_sam_equals_reference_with_its_own_mask(a, never, hg38["got"]["mh"])
    assert masked == len(never)


def test_configs3_pacbio_20000_reads_identical(hg38):
    assert hg38["stats"]["long"].total_reads == N_LONG
    got, want = hg38["got"]["long"], hg38["want"]["long"]
    if got != want:
        g, w = got.split(b"\n"), want.split(b"\n")
        bad = [i for i, (x, y) in enumerate(zip(g, w)) if x != y]
        assert False, "%d / %d lines, %d differ, first: %r ... vs %r ..." % (len(g), len(w), len(bad), g[bad[0]][:300] if bad else None, w[bad[0]][:300] if bad else None)


def test_a_session_does_not_carry_the_unset_flag_into_the_next_run(hg38):
    """KART_AMD_UNSET_FLAG was only ever written, so a later run of the same session still printed the sentinel.
    The -pacbio and default runs above ran in the session before / after the -m run: neither may hold the sentinel."""
    needle = b"\t%d\t" % UNSET_FLAG
    assert needle not in hg38["got"]["pe"] and needle not in hg38["got"]["long"]

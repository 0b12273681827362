"""CPU: ordinary gzip input inflated by several threads (kart_amd/csrc/host/detail/pgzip.inc) -- the reference reads a .gz library
through gzgets() on one thread (src/GetData.cpp:145-219).  Two levels:
* tests/cpu_backend/pgz_check.cpp holds the reader's bytes against zlib's on the same file: every delivered byte equal, a finished
  file delivered whole, a damaged or truncated one delivered as a PREFIX of what zlib still inflates (the caller continues with zlib
  from there, so the reads the reference still sees are the reads mapped);
* the host pipeline (bound to the CPU oracle backend) maps gz libraries through it and must write the golden SAM -- chunk sizes of
  a few KB so that block search, chunk chaining, window resolution, several rounds and several members all occur on the small
  fixtures."""
import gzip
import json
import os
import random
import subprocess
import zlib

import pytest

from conftest import GOLDEN, ROOT, SMALL_PREFIX

SAM = os.path.join(GOLDEN, "sam")


@pytest.fixture(scope="module")
def pgz_check():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpu_backend"), os.path.join("..", "..", "tests", "_build", "pgz_check"),
                           os.path.join("..", "..", "tests", "_build", "pgz_check-san")], stdout=subprocess.DEVNULL)
    return os.path.join(ROOT, "tests", "_build", "pgz_check")


@pytest.fixture(scope="module")
def host_oracle_binary():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpu_backend")], stdout=subprocess.DEVNULL)
    return os.path.join(ROOT, "tests", "_build", "kart-host-oracle")


def check(binary, tmp_path, data, threads=4, chunk_kb=8):
    path = str(tmp_path / "t.gz")
    open(path, "wb").write(data)
    r = subprocess.run([binary, path, str(threads), str(chunk_kb)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       env=dict(os.environ, KART_AMD_PGZ_MIN_KB="0", ASAN_OPTIONS="detect_leaks=0"))
    assert r.returncode in (0, 1), r.stderr.decode()[-600:]
    line = json.loads(r.stdout)
    assert r.returncode == 0 and line["equal"] == 1, line
    return line


def fastq_text():
    return gzip.open(os.path.join(SAM, "pe_1.fq.gz")).read()


def test_every_byte_is_zlibs_byte(pgz_check, tmp_path):
    """compression levels (block sizes and match lengths differ), stored and fixed-code blocks, flush points, several members, a header
    with a file name, 2-8 threads, chunks from 4 KB (smaller than a block: most searches find nothing and the chunk before runs through)
    to 256 KB (one round)"""
    text = fastq_text()
    whole = 0
    for level in (1, 4, 6, 9):
        for threads, chunk in ((2, 4), (4, 8), (8, 32), (3, 256)):
            line = check(pgz_check, tmp_path, gzip.compress(text, level), threads, chunk)
            assert line["opened"] == 1 and line["all_done"] == 1 and line["bytes"] == len(text), line
            whole += 1
    assert whole == 16
    co = zlib.compressobj(6, zlib.DEFLATED, 31, 9, zlib.Z_FIXED)
    fixed = co.compress(text) + co.flush()
    co = zlib.compressobj(6, zlib.DEFLATED, 31)
    flushed = b"".join(co.compress(text[i:i + 50000]) + co.flush(zlib.Z_FULL_FLUSH if i % 100000 else zlib.Z_SYNC_FLUSH) for i in range(0, len(text), 50000)) + co.flush()
    named = str(tmp_path / "named.gz")
    with gzip.GzipFile(filename="reads_1.fq", mode="wb", fileobj=open(named, "wb")) as fh:
        fh.write(text)
    third = len(text) // 3
    for data in (gzip.compress(text, 0), fixed, flushed, open(named, "rb").read(),
                 gzip.compress(text[:third]) + gzip.compress(text[third:2 * third], 1) + gzip.compress(text[2 * third:], 9),
                 gzip.compress(text) + gzip.compress(b"")):
        line = check(pgz_check, tmp_path, data, 4, 8)
        assert line["all_done"] == 1 and line["bytes"] == len(text), line
    # the sanitizer build on two of them (ASan / UBSan run on the CPU build only)
    for data in (gzip.compress(text, 6), flushed):
        line = check(pgz_check + "-san", tmp_path, data, 4, 8)
        assert line["all_done"] == 1 and line["bytes"] == len(text), line


def test_what_it_declines_it_leaves_to_zlib(pgz_check, tmp_path):
    """not a gzip file, one thread, text that is not text (the block search's test would not hold), a member followed by bytes that are
    no member: nothing or a prefix is delivered, never a wrong byte"""
    text = fastq_text()
    line = check(pgz_check, tmp_path, gzip.compress(text), 1, 8)
    assert line["opened"] == 0 and line["bytes"] == 0
    line = check(pgz_check, tmp_path, gzip.compress(os.urandom(100000) + text), 4, 8)
    assert line["bytes"] == 0 and line["all_done"] == 0
    line = check(pgz_check, tmp_path, gzip.compress(text) + b"trailing bytes that are no gzip member at all", 4, 8)
    assert line["all_done"] == 0 and line["bytes"] <= len(text)
    line = check(pgz_check, tmp_path, gzip.compress(text[:200000] + bytes([200, 201, 202]) + text[200000:]), 4, 8)
    assert line["all_done"] == 0 and line["bytes"] <= 200000


def test_damaged_and_truncated_streams_yield_a_prefix(pgz_check, tmp_path):
    """a flipped bit anywhere, a file cut anywhere: what the reader delivers is a prefix of what zlib still inflates (check() asserts it);
    and it is most of it when the damage is late"""
    text = fastq_text()
    data = gzip.compress(text)
    rng = random.Random(7)
    for _ in range(30):
        bad = bytearray(data)
        bad[rng.randrange(12, len(bad))] ^= 1 << rng.randrange(8)
        check(pgz_check, tmp_path, bytes(bad), rng.choice((2, 4, 8)), rng.choice((4, 8, 16, 64)))
    for _ in range(10):
        check(pgz_check, tmp_path, data[: rng.randrange(100, len(data))], 4, rng.choice((4, 8, 64)))
    line = check(pgz_check, tmp_path, data[: len(data) - 1000], 4, 8)
    assert line["bytes"] > len(text) // 2, line


def test_gz_libraries_through_the_parallel_reader_give_the_golden_sam(host_oracle_binary, tmp_path):
    """mate files, an interleaved file, -m, single-end; chunk sizes that put several rounds into one batch and several batches into one
    round; a two-member file; and the switch that turns the reader off"""
    r1, r2 = fastq_text(), gzip.open(os.path.join(SAM, "pe_2.fq.gz")).read()
    want = gzip.open(os.path.join(SAM, "pe.sam.gz")).read()

    def put(name, data):
        path = str(tmp_path / name)
        open(path, "wb").write(data)
        return path

    def run(args, env):
        out = str(tmp_path / "o.sam")
        r = subprocess.run([host_oracle_binary, "-silent", "-i", SMALL_PREFIX] + args + ["-t", "8", "-o", out], stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, env=dict(os.environ, KART_AMD_PGZ_MIN_KB="0", **env))
        assert r.returncode == 0, r.stderr.decode()[-400:]
        return open(out, "rb").read(), r.stderr.decode()

    g1, g2 = put("p1.fq.gz", gzip.compress(r1)), put("p2.fq.gz", gzip.compress(r2, 9))
    for chunk in ("4", "16", "256"):
        got, log = run(["-f", g1, "-f2", g2], {"KART_AMD_PGZ_CHUNK_KB": chunk, "KART_AMD_PGZ_DEBUG": "1"})
        assert got == want, chunk
        assert "pgz round" in log                                   # (the reader did run)
    got, log = run(["-f", g1, "-f2", g2], {"KART_AMD_NO_PGZ": "1", "KART_AMD_PGZ_DEBUG": "1"})
    assert got == want and "pgz round" not in log
    half = len(r1) // 2
    assert run(["-f", put("m1.fq.gz", gzip.compress(r1[:half]) + gzip.compress(r1[half:])), "-f2", g2], {"KART_AMD_PGZ_CHUNK_KB": "8"})[0] == want
    assert run(["-f", g1, "-f2", g2, "-m"], {"KART_AMD_PGZ_CHUNK_KB": "8"})[0] == gzip.open(os.path.join(SAM, "pe_m.sam.gz")).read()
    inter = gzip.open(os.path.join(SAM, "pe_interleaved.fq.gz")).read()
    assert run(["-f", put("i.fq.gz", gzip.compress(inter)), "-p"], {"KART_AMD_PGZ_CHUNK_KB": "8"})[0] == gzip.open(os.path.join(SAM, "pe_interleaved.sam.gz")).read()
    assert run(["-f", os.path.join(SAM, "se.fq.gz")], {"KART_AMD_PGZ_CHUNK_KB": "8"})[0] == gzip.open(os.path.join(SAM, "se.sam.gz")).read()


def test_a_damaged_gz_library_through_the_parallel_reader_matches_the_live_reference(host_oracle_binary, tmp_path):
    """the reads in front of the damage are the reads the reference's gzgets() loop still sees (as in
    tests/test_host_pipeline.py::test_damaged_gz_input_yields_what_the_reference_still_reads, with the several-thread reader in front of zlib)"""
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "kart")
    if not os.path.exists(ref_bin):
        pytest.skip("oracle/_ref/kart not present")
    r1, r2 = fastq_text(), gzip.open(os.path.join(SAM, "pe_2.fq.gz")).read()
    good2 = str(tmp_path / "g2.fq.gz")
    open(good2, "wb").write(gzip.compress(r2))
    for where in (0.5, 0.9):
        d = bytearray(gzip.compress(r1))
        d[int(len(d) * where)] ^= 0x55
        bad1 = str(tmp_path / "bad_1.fq.gz")
        open(bad1, "wb").write(bytes(d))
        outs = []
        for binary, t in ((ref_bin, "1"), (host_oracle_binary, "8")):
            out = str(tmp_path / "o.sam")
            r = subprocess.run([binary, "-silent", "-i", SMALL_PREFIX, "-f", bad1, "-f2", good2, "-t", t, "-o", out], stdout=subprocess.DEVNULL,
                               stderr=subprocess.DEVNULL, env=dict(os.environ, KART_AMD_PGZ_MIN_KB="0", KART_AMD_PGZ_CHUNK_KB="8"))
            assert r.returncode == 0 or binary == ref_bin
            outs.append(open(out, "rb").read().split(b"\n") if r.returncode == 0 else None)
        ref, got = outs
        assert len(got) > 1000
        if ref is None:
            continue
        assert len(got) == len(ref), (where, len(got), len(ref))
        differing = [i for i, (x, y) in enumerate(zip(ref, got)) if x != y]
        assert len(differing) <= 1 and all(i >= len(ref) - 4 for i in differing), (where, differing[:5])


def test_the_benchs_gzip_writer_writes_one_ordinary_member(pgz_check, tmp_path):
    """benchkit/gz.py (slices deflated side by side, one header, one trailer): gzip reads the text back, zlib sees ONE member, and the
    several-thread reader takes it like any other file"""
    import sys
    sys.path.insert(0, ROOT)
    from benchkit.gz import gzip_one_member
    text = fastq_text()
    src, dst = str(tmp_path / "t.fq"), str(tmp_path / "t.fq.gz")
    for data, slice_bytes in ((text, 100000), (text, 8 << 20), (b"", 4096), (text[:4096], 4096)):
        open(src, "wb").write(data)
        assert gzip_one_member(src, dst, 6, 4, slice_bytes) == (len(data), os.path.getsize(dst))
        packed = open(dst, "rb").read()
        assert gzip.decompress(packed) == data
        d = zlib.decompressobj(31)
        assert d.decompress(packed) == data and d.eof and d.unused_data == b""          # one member, nothing behind it
    open(src, "wb").write(text)
    gzip_one_member(src, dst, 6, 4, 100000)
    line = check(pgz_check, tmp_path, open(dst, "rb").read(), 4, 16)
    assert line["all_done"] == 1 and line["bytes"] == len(text)


def test_a_gz_library_on_its_way_to_the_device_stream(tmp_path):
    """GzText::fill_direct + GzProducer (kart_amd/csrc/host/detail/batch_reader.inc) without a device -- tests/cpu_backend/gzproducer_check.cpp plays the
    stream's part: blocks of the growing text against zlib's, a look-ahead small enough that the inflating thread waits for the consumer's releases,
    and the hand-over half-way that Source::gz_stream_end makes (the rest through the gz reader's carry and fill()).  Ordinary gzip through the
    several-thread reader (rounds straight into the block), several members, BGZF, one zlib stream, a damaged stream; also under ThreadSanitizer
    and AddressSanitizer / UBSan."""
    from bgzf_util import bgzf
    targets = [os.path.join("..", "..", "tests", "_build", "gzproducer_check" + s) for s in ("", "-tsan", "-asan")]
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "cpu_backend")] + targets, stdout=subprocess.DEVNULL)
    text = fastq_text()

    def run(data, suffix="", threads=4, block_kb=100, ahead_mb=64, env=None):
        path = str(tmp_path / "t.gz")
        open(path, "wb").write(data)
        r = subprocess.run([os.path.join(ROOT, "tests", "_build", "gzproducer_check" + suffix), path, str(threads), str(block_kb), str(ahead_mb)],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                           env=dict(os.environ, KART_AMD_PGZ_MIN_KB="0", KART_AMD_PGZ_CHUNK_KB="16", ASAN_OPTIONS="detect_leaks=0", **(env or {})))
        assert r.returncode == 0 and b"ThreadSanitizer" not in r.stderr and b"runtime error" not in r.stderr, (r.stdout, r.stderr[-800:])
        line = json.loads(r.stdout)
        assert line["equal"] == 1
        return line

    plain = gzip.compress(text)
    for suffix in ("", "-tsan", "-asan"):
        assert run(plain, suffix)["through_the_producer"] == len(text)
        assert run(bgzf(text), suffix)["through_the_producer"] == len(text)
    third = len(text) // 3
    assert run(gzip.compress(text[:third]) + gzip.compress(text[third:], 1), block_kb=37)["with_a_hand_over_half_way"] == len(text)
    assert run(plain, env={"KART_AMD_NO_PGZ": "1"}, block_kb=700)["through_the_producer"] == len(text)
    assert run(plain, threads=8, block_kb=3)["through_the_producer"] == len(text)
    damaged = bytearray(plain)
    damaged[len(damaged) * 6 // 10] ^= 0x55
    line = run(bytes(damaged), "-tsan")                       # (equal: the producer's text is what small zlib reads still deliver of it)
    assert 0 < line["through_the_producer"] < len(text)

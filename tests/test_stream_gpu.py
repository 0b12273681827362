"""GPU: the device's FASTQ parser and SAM formatter (kg_stream_*, kart_amd/csrc/stream_kernels.hip) against the reference's own
arithmetic -- GetNextEntry / GetNextChunk (reference src/GetData.cpp:29-143) restated in a few lines of Python below -- and the
product binary through that path against the golden SAM of the unmodified reference."""
import gzip
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, SMALL_PREFIX

pytestmark = pytest.mark.gpu
KART_AMD = os.path.join(ROOT, "kart_amd", "bin", "kart-amd")


def comp(c):    # GetComplementaryBase, src/tools.cpp:3-17
    return {65: 84, 97: 84, 67: 71, 99: 71, 71: 67, 103: 67, 84: 65, 116: 65}.get(c, 78)


def reference_reads(text: bytes):
    """GetNextEntry over a whole file: (name, sequence, qualities) per record; the last character of every line is taken to be
    the newline (src/GetData.cpp:66-69); stops like getline() at the end of the text"""
    lines = text.split(b"\n")
    ends_nl = text.endswith(b"\n")
    if ends_nl:
        lines = lines[:-1]
    full = [l + b"\n" for l in lines]
    if not ends_nl and full:
        full[-1] = full[-1][:-1]          # the last line has no newline
    out = []
    for i in range(0, len(full) - 3, 4):
        h, s, q = full[i], full[i + 1], full[i + 3]
        p1 = p2 = len(h) - 1
        for k in range(1, len(h)):
            if h[k] not in b">@":
                p1 = k
                break
        for k in range(1, len(h)):
            if h[k] in b" /\t":
                p2 = k
                break
        name = h[p1:p2] if p2 > p1 else b""
        rlen = len(s) - 1
        out.append((name, s[:rlen], q[:min(len(q), rlen)]))
    return out


def held(seq: bytes, flip: bool) -> bytes:
    return bytes(comp(c) for c in reversed(seq)) if flip else seq


@pytest.fixture(scope="module")
def stream(gpu_index_full):
    from kart_amd import api
    s = api.Stream(gpu_index_full, max_reads=16000, max_window=8 << 20, lanes=2)
    yield s
    s.close()


def _golden_text(name):
    return gzip.open(os.path.join(GOLDEN, "sam", name + ".gz")).read()


def test_parser_two_files_equals_getnextchunk(stream):
    t1, t2 = _golden_text("pe_1.fq"), _golden_text("pe_2.fq")
    r1, r2 = reference_reads(t1), reference_reads(t2)
    p = stream.parse(t1, t2, paired=True, want_reads=16000)
    assert (p.n_reads, p.n_chunks, p.stop, p.done) == (2 * len(r1), 3, 0, 1) and p.used[0] == len(t1) and p.used[1] == len(t2)
    reads = stream.fetch_reads(p)
    for i, got in enumerate(reads):
        src = (r1, r2)[i & 1][i >> 1]
        assert got == held(src[1], bool(i & 1)), i


def test_parser_takes_whole_chunks_and_reports_the_next_window(stream):
    t1, t2 = _golden_text("pe_1.fq"), _golden_text("pe_2.fq")
    r1 = reference_reads(t1)
    # not at the end of the file: only whole chunks of 4000 reads (2000 records per file), the rest stays for the next window
    p = stream.parse(t1, t2, paired=True, want_reads=16000, eof=(False, False), begin=(1000, 77))
    assert (p.n_reads, p.n_chunks, p.stop, p.done) == (8000, 2, 0, 0)
    lines = t1.split(b"\n")
    assert p.used[0] == 1000 + len(b"\n".join(lines[:16000])) + 1          # 4000 records of four lines
    reads = stream.fetch_reads(p)
    assert len(reads) == 8000 and reads[0] == r1[0][1] and reads[7998] == r1[3999][1]
    # a window cut in the middle of a record: the partial record is not taken
    cut = len(b"\n".join(lines[:8002])) + 5
    p = stream.parse(t1[:cut], t2, paired=True, want_reads=16000, eof=(False, False))
    assert (p.n_reads, p.stop, p.done) == (4000, 0, 0)


def test_parser_edge_cases(stream):
    def fq(recs, last_newline=True):
        t = b"".join(b"@" + n + b"\n" + s + b"\n+\n" + q + b"\n" for n, s, q in recs)
        return t if last_newline else t[:-1]
    recs = [(b"r%d extra words/1" % i, b"ACGTNacgtnRYKM"[: 5 + i % 9] * 3, b"I" * (3 * (5 + i % 9))) for i in range(40)]
    # one file, interleaved pairs, lower case and ambiguity codes through the reverse complement, header cut at ' ' '/' '\t'
    text = fq(recs)
    p = stream.parse(text, None, paired=True, chunk_reads=8, want_reads=40)
    assert (p.n_reads, p.n_chunks, p.stop, p.done) == (40, 5, 0, 1)
    want = reference_reads(text)
    for i, got in enumerate(stream.fetch_reads(p)):
        assert got == held(want[i][1], bool(i & 1)), i
    # no newline behind the last line: the reference drops that line's last character only for the sequence line; the record counts
    p = stream.parse(fq(recs, last_newline=False), None, paired=False, chunk_reads=8, want_reads=40)
    assert (p.n_reads, p.done) == (40, 1)
    # an empty read ends a chunk early in the reference: the device stops in front of its chunk and says so
    bad = list(recs)
    bad[19] = (b"empty", b"", b"")
    p = stream.parse(fq(bad), None, paired=False, chunk_reads=8, want_reads=40)
    assert (p.n_reads, p.stop, p.done) == (16, 1, 0)
    # a lone mate at the end of an interleaved file: whole chunks only, the tail is the host reader's
    p = stream.parse(fq(recs[:39]), None, paired=True, chunk_reads=8, want_reads=40)
    assert (p.n_reads, p.stop, p.done) == (32, 2, 0)
    # mate files of different length
    p = stream.parse(fq(recs[:20]), fq(recs[:17]), paired=True, chunk_reads=8, want_reads=40)
    assert (p.n_reads, p.stop, p.done) == (32, 2, 0)
    # a NUL byte: lines are C strings in the reference
    p = stream.parse(fq(recs).replace(b"r7 ", b"r7\0"), None, paired=False, chunk_reads=8, want_reads=40)
    assert (p.n_reads, p.stop) == (0, 1)
    # CR LF line ends: the '\r' belongs to the sequence (rlen counts it), as in the reference
    text = fq(recs).replace(b"\n", b"\r\n")
    p = stream.parse(text, None, paired=False, chunk_reads=8, want_reads=40)
    want = reference_reads(text)
    got = stream.fetch_reads(p)
    assert p.n_reads == 40 and all(g == w[1] for g, w in zip(got, want))


def test_stream_text_equals_the_reference_sam(stream):
    """the text the device prints for the golden paired-end set == the lines of the unmodified reference's SAM (reads handed back
    to the host print nothing here)"""
    t1, t2 = _golden_text("pe_1.fq"), _golden_text("pe_2.fq")
    want = [l + b"\n" for l in _golden_text("pe.sam").split(b"\n") if l and not l.startswith(b"@")]
    p = stream.parse(t1, t2, paired=True, want_reads=4000, lane=1)
    assert p.n_reads == 4000
    text, host = stream.map(lane=1)          # EstDistance = MaxInsertSize for the first chunk, as in the reference
    assert len(host) < 200 and len(host) % 2 == 0
    decided = [i for i in range(4000) if i not in set(host)]
    # default mode: one line per read
    assert all(text[i].count(b"\n") == 1 for i in decided) and all(text[i] == b"" for i in host)
    assert [text[i] for i in decided] == [want[i] for i in decided]
    t = stream.timing()
    assert t["batches"] >= 1 and t["search_kernel_ms"] > 0 and t["text_out_bytes"] > 0


def _run(args, out, env=None):
    r = subprocess.run([KART_AMD, "-silent", "-i", SMALL_PREFIX, "-o", out, "-t", "8"] + args, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       env=dict(os.environ, KART_AMD_VERBOSE="1", **(env or {})))
    assert r.returncode == 0, r.stdout.decode()[-600:]
    return r.stdout.decode()


@pytest.mark.parametrize("case,extra", [("pe", []), ("pe_m", ["-m"]), ("se", []), ("pe_interleaved", ["-p"])])
def test_product_runs_through_the_stream(case, extra, built_lib, tmp_path):
    files = {"pe": ("pe_1.fq", "pe_2.fq"), "pe_m": ("pe_1.fq", "pe_2.fq"), "se": ("se.fq",), "pe_interleaved": ("pe_interleaved.fq",)}[case]
    paths = []
    for f in files:
        dst = str(tmp_path / f)
        open(dst, "wb").write(_golden_text(f))
        paths.append(dst)
    args = ["-f", paths[0]] + (["-f2", paths[1]] if len(paths) > 1 else []) + extra
    out = str(tmp_path / "o.sam")
    log = _run(args, out, {"KART_AMD_UNSET_FLAG": "0"})
    assert "device stream:" in log, log[-400:]
    want = _golden_text(case + ".sam")
    got = open(out, "rb").read()
    assert got == want
    # the same input through the host's own reader and printer
    out2 = str(tmp_path / "o2.sam")
    log2 = _run(args, out2, {"KART_AMD_NO_STREAM": "1", "KART_AMD_UNSET_FLAG": "0"})
    assert "device stream:" not in log2
    assert open(out2, "rb").read() == got


def test_grouped_seeding_one_launch_for_several_lanes(gpu_index_full):
    """seed_group: the lanes of a group seed their batches in ONE launch per round (kg_stream_map returns when the round is done).
    Every lane's text must be what the lane prints on its own -- with all lanes present, with a lane absent for a round, with
    different batches in the lanes, and round after round."""
    import threading
    from kart_amd import api
    t1, t2 = _golden_text("pe_1.fq"), _golden_text("pe_2.fq")

    def cut(text, r0, r1):            # records [r0, r1) of a 4-line FASTQ text
        lines = text.split(b"\n")
        return b"\n".join(lines[4 * r0:4 * r1]) + b"\n"

    batches = [(cut(t1, 0, 2000), cut(t2, 0, 2000)), (cut(t1, 2000, 4000), cut(t2, 2000, 4000)), (cut(t1, 4000, 4500), cut(t2, 4000, 4500))]
    solo = api.Stream(gpu_index_full, max_reads=8000, max_window=4 << 20, lanes=1)
    want = []
    for a, b in batches:
        p = solo.parse(a, b, paired=True, want_reads=4000)
        want.append(solo.map())
    single_launches = solo.timing()["search_kernel_launches"]
    solo.close()
    assert single_launches == 3
    s = api.Stream(gpu_index_full, max_reads=8000, max_window=4 << 20, lanes=4, seed_group=2)
    try:
        def round_of(assign):         # {lane: batch index}: parse in the calling thread, map from one thread per lane
            got, errs = {}, []
            for lane, bi in assign.items():
                p = s.parse(batches[bi][0], batches[bi][1], paired=True, want_reads=4000, lane=lane)
                assert p.n_reads == 2 * (batches[bi][0].count(b"\n") // 4)

            def work(lane):
                try:
                    got[lane] = s.map(lane=lane)
                except Exception as exc:      # noqa: BLE001
                    errs.append((lane, exc))
            th = [threading.Thread(target=work, args=(lane,)) for lane in assign]
            for t in th:
                t.start()
            for t in th:
                t.join(120)
            assert not errs, errs
            assert all(not t.is_alive() for t in th), "a lane never came back from its group's round"
            for lane, bi in assign.items():
                assert got[lane] == want[bi], (lane, bi)
        for lane in range(4):
            s.group_absent(lane, 0)
        s.timing(reset=True)
        round_of({0: 0, 1: 1})                      # group 0 complete: one launch for two batches
        assert s.timing()["search_kernel_launches"] == 1
        round_of({0: 2, 1: 0, 2: 1, 3: 1})          # both groups, different batch sizes in the lanes
        assert s.timing()["search_kernel_launches"] == 3
        s.group_absent(1, 1)                        # lane 1 sits this round out: lane 0's round completes without it
        round_of({0: 1})
        round_of({0: 0, 1: 2})                      # ... and it is back in the next
        s.group_absent(3, -1)                       # lane 3's input has ended: lane 2 goes on alone, round after round
        round_of({2: 0})
        round_of({2: 2})
        # a lane that is absent until further notice may not arrive with a batch
        s.parse(batches[0][0], batches[0][1], paired=True, want_reads=4000, lane=3)
        with pytest.raises(api.KartAmdError):
            s.map(lane=3)
        s.group_absent(3, 0)
        round_of({2: 1, 3: 0})
        # a caller's failure path: a lane that waits for its group comes back with an error once the groups are aborted ...
        s.parse(batches[0][0], batches[0][1], paired=True, want_reads=4000, lane=0)
        errs = []

        def waits():
            try:
                s.map(lane=0)
            except api.KartAmdError as exc:
                errs.append(str(exc))
        t = threading.Thread(target=waits)
        t.start()
        import time
        time.sleep(0.3)
        assert t.is_alive()                          # (lane 1 has neither arrived nor said that it has no batch)
        s.group_abort()
        t.join(30)
        assert not t.is_alive() and errs and "aborted" in errs[0]
        # ... and the next run starts clean
        for lane in range(4):
            s.group_absent(lane, 0)
        round_of({0: 1, 1: 0})
    finally:
        s.close()


@pytest.mark.parametrize("lanes,reads,group", [(1, 4000, 0), (2, 8000, 0), (4, 4000, 0), (4, 4000, 4), (8, 4000, 4), (4, 4000, 2), (6, 8000, 3), (8, 1 << 20, 4)])
def test_stream_with_small_batches_and_many_lanes(lanes, reads, group, built_lib, tmp_path):
    """batches of one or two chunks through 1..8 lanes, every lane seeding for itself or in groups of 2..4 (partial groups where the
    input ends): every carry / ordering path of the feeder, byte-identical output"""
    paths = []
    for f in ("pe_1.fq", "pe_2.fq"):
        dst = str(tmp_path / f)
        open(dst, "wb").write(_golden_text(f))
        paths.append(dst)
    out = str(tmp_path / "o.sam")
    log = _run(["-f", paths[0], "-f2", paths[1]], out, {"KART_AMD_STREAM_LANES": str(lanes), "KART_AMD_STREAM_READS": str(reads), "KART_AMD_SEED_GROUP": str(group)})
    assert "device stream:" in log
    assert open(out, "rb").read() == _golden_text("pe.sam")
    if lanes == 8 and group == 4:
        # the same configuration again and again (a null-stream memset once raced with the lanes' first batches: 4-11 of 16 such
        # runs came out different), and with the writer's pwrite thread in the mix
        for i in range(5):
            env = {"KART_AMD_STREAM_LANES": str(lanes), "KART_AMD_STREAM_READS": str(reads), "KART_AMD_SEED_GROUP": str(group)}
            if i % 2:
                env.update({"KART_AMD_WRITER_THREADS": "2", "KART_AMD_PWRITE_THREADS": "1"})
            _run(["-f", paths[0], "-f2", paths[1]], out, env)
            assert open(out, "rb").read() == _golden_text("pe.sam"), i

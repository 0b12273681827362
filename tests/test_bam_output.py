"""-bo: the BAM file must decode (an independent reader of the public format, below) to exactly the records the same run prints
as SAM with -o.  The reference makes its BAM by feeding each printed SAM line to htslib's parser (src/Mapping.cpp:610-620);
htslib cannot be built here, so the check is against the format, not against bytes of the reference."""
import gzip
import os
import struct
import subprocess

import pytest

from conftest import ROOT, SMALL_PREFIX
from test_host_pipeline import CASES, host_oracle_binary, materialise  # noqa: F401  (fixture)

NT16 = "=ACMGRSVTWYHKDBN"


def decode_bam(path):
    """-> (header text, [sam-like field tuples]) via gzip's multi-member reader (BGZF blocks are gzip members)"""
    raw = gzip.open(path, "rb").read()
    assert raw[:4] == b"BAM\x01"
    l_text, = struct.unpack_from("<i", raw, 4)
    text = raw[8:8 + l_text].decode()
    at = 8 + l_text
    n_ref, = struct.unpack_from("<i", raw, at); at += 4
    refs = []
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<i", raw, at); at += 4
        name = raw[at:at + l_name - 1].decode(); at += l_name
        l_ref, = struct.unpack_from("<i", raw, at); at += 4
        refs.append((name, l_ref))
    recs = []
    while at < len(raw):
        block, = struct.unpack_from("<i", raw, at); at += 4
        rid, pos, l_qn, mapq, bin_, n_cig, flag, l_seq, rnext, pnext, tlen = struct.unpack_from("<iiBBHHHiiii", raw, at)
        p = at + 32
        qname = raw[p:p + l_qn - 1].decode(); p += l_qn
        cig = ""
        ref_len = 0
        for _ in range(n_cig):
            v, = struct.unpack_from("<I", raw, p); p += 4
            cig += "%d%s" % (v >> 4, "MIDNSHP=X"[v & 15])
            if (v & 15) in (0, 2, 3, 7, 8):
                ref_len += v >> 4
        seq = "".join(NT16[raw[p + i // 2] >> 4 if i % 2 == 0 else raw[p + i // 2] & 15] for i in range(l_seq)); p += (l_seq + 1) // 2
        q = raw[p:p + l_seq]; p += l_seq
        qual = "*" if l_seq == 0 or q == b"\xff" * l_seq else "".join(chr(c + 33) for c in q)
        tags = []
        while p < at + block:
            tag = raw[p:p + 2].decode(); ty = chr(raw[p + 2]); p += 3
            fmt = {"c": "<b", "C": "<B", "s": "<h", "S": "<H", "i": "<i", "I": "<I"}[ty]
            v, = struct.unpack_from(fmt, raw, p); p += struct.calcsize(fmt)
            assert (ty == "C" and 0 <= v < 256) or (ty == "S" and 256 <= v < 65536) or (ty == "I" and v >= 65536) or (ty == "c" and -128 <= v < 0) or \
                   (ty == "s" and -32768 <= v < -128) or (ty == "i" and v < -32768), (tag, ty, v)       # smallest type that holds the value
            tags.append("%s:i:%d" % (tag, v))
        assert p == at + block
        end = pos + (ref_len if ref_len else 1)
        assert bin_ == reg2bin(pos, end), (qname, bin_)
        rname = refs[rid][0] if rid >= 0 else "*"
        rn = "*" if rnext < 0 else ("=" if rnext == rid else refs[rnext][0])
        recs.append((qname, flag, rname, pos + 1, mapq, cig or "*", rn, pnext + 1, tlen, seq or "*", qual) + tuple(tags))
        at += block
    return text, refs, recs


def reg2bin(beg, end):
    end -= 1
    for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return base + (beg >> shift)
    return 0


def sam_records(path):
    head, recs = "", []
    for line in open(path, "rb").read().decode().splitlines():
        if line.startswith("@"):
            head += line + "\n"
            continue
        f = line.split("\t")
        seq = "".join(c.upper() if c.upper() in NT16 else "N" for c in f[9]) if f[9] != "*" else "*"
        recs.append((f[0], int(f[1]), f[2], int(f[3]), int(f[4]), f[5], f[6], int(f[7]), int(f[8]), seq, f[10]) + tuple(f[11:]))
    return head, recs


@pytest.mark.parametrize("case", ["pe", "pe_m", "se_fasta", "edge_pe", "pacbio", "edge_multi_lib"])
def test_bam_equals_sam(case, host_oracle_binary, tmp_path):
    args = [materialise(str(tmp_path), a) if a.endswith((".fq", ".fa", ".gz")) else a for a in CASES[case]]
    sam, bam = str(tmp_path / "o.sam"), str(tmp_path / "o.bam")
    for flag, out in (("-o", sam), ("-bo", bam)):
        r = subprocess.run([host_oracle_binary, "-silent", "-t", "3", "-i", SMALL_PREFIX] + args + [flag, out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        assert r.returncode == 0, r.stdout.decode()[-400:]
    head, want = sam_records(sam)
    text, refs, got = decode_bam(bam)
    assert text == head
    assert [n for n, _ in refs] == [l.split("\t")[1][3:] for l in head.splitlines() if l.startswith("@SQ")]
    assert len(got) == len(want)
    for g, w in zip(got, want):
        assert g == w
    # the file ends with the 28-byte empty BGZF block
    assert open(bam, "rb").read()[-28:] == bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")

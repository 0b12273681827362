"""BGZF (bgzip) writer for the tests: a series of gzip members of at most 64 KB of text, each with its compressed size in a 'BC'
extra field (SAMv1 4.1), ending with the empty member bgzip appends."""
import struct, zlib, random
def bgzf_block(data: bytes, level: int = 6) -> bytes:
    c = zlib.compressobj(level, zlib.DEFLATED, -15)
    body = c.compress(data) + c.flush()
    bsize = 12 + 6 + len(body) + 8
    hdr = b"\x1f\x8b\x08\x04" + b"\x00\x00\x00\x00" + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1)
    return hdr + body + struct.pack("<II", zlib.crc32(data) & 0xffffffff, len(data) & 0xffffffff)
EOF_BLOCK = bgzf_block(b"")
def bgzf(data: bytes, block=0xff00, rng=None, eof=True, level: int = 6) -> bytes:
    out = []; at = 0
    while at < len(data):
        n = block if rng is None else rng.randint(1, block)
        out.append(bgzf_block(data[at:at + n], level)); at += n
    if eof: out.append(EOF_BLOCK)
    return b"".join(out)

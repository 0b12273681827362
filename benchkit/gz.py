"""benchkit/gz.py -- BENCH / TEST INFRASTRUCTURE: an ordinary single-member gzip file written by several threads.

What `gzip file.fq` writes is one deflate stream; the bench's gz leg needs such files of a GB each within seconds.  As pigz does
without its shared dictionary: slices of the text are deflated independently (raw deflate, each ended on a byte boundary by a sync
flush's empty stored block, the last with the final-block bit), laid end to end behind one gzip header, and closed with the CRC-32
and length of the whole text.  Any inflater reads it as one member; a member-parallel reader (BGZF) finds nothing to split."""
import os
import zlib
from concurrent.futures import ThreadPoolExecutor


def gzip_one_member(src, dst, level=6, threads=8, slice_bytes=8 << 20):
    """-> (text bytes, compressed bytes).  zlib releases the interpreter lock inside compress(), so the threads do run side by side."""
    size = os.path.getsize(src)
    n = max(1, (size + slice_bytes - 1) // slice_bytes)
    fd = os.open(src, os.O_RDONLY)

    def piece(i):
        buf = os.pread(fd, slice_bytes, i * slice_bytes)
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        return c.compress(buf) + c.flush(zlib.Z_FINISH if i == n - 1 else zlib.Z_SYNC_FLUSH)

    def whole_crc():
        crc, at = 0, 0
        while at < size:
            buf = os.pread(fd, 64 << 20, at)
            crc = zlib.crc32(buf, crc)
            at += len(buf)
        return crc

    try:
        with ThreadPoolExecutor(max(1, threads)) as ex, open(dst, "wb") as out:
            crc = ex.submit(whole_crc)
            out.write(b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\x03")
            for part in ex.map(piece, range(n)):
                out.write(part)
            out.write((crc.result() & 0xFFFFFFFF).to_bytes(4, "little") + (size & 0xFFFFFFFF).to_bytes(4, "little"))
    finally:
        os.close(fd)
    return size, os.path.getsize(dst)

#!/usr/bin/env python3
"""tools/fuzz_bgzf.py [cases] -- mutated bgzip-ped FASTQ through the gz reader under AddressSanitizer / UBSan (CPU build:
`make -C tests/cpu_backend asan`; KART_FUZZ_BIN overrides the binary): random bytes, header fields of a member (FLG, XLEN, the BC
subfield, BSIZE), truncation, a member's CRC / ISIZE.  The run must end with status 0 (whatever it could still read is mapped) or 1
(an error message), never with a sanitizer report or a signal."""
import gzip
import os
import random
import struct
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from bgzf_util import bgzf  # noqa: E402

BIN = os.environ.get("KART_FUZZ_BIN", os.path.join(ROOT, "tests", "_build", "kart-host-oracle-asan"))
SMALL = os.path.join(ROOT, "tests", "golden", "idx", "small")
SAM = os.path.join(ROOT, "tests", "golden", "sam")


def members(d):
    pos, out = 0, []
    while pos < len(d) - 18:
        n = struct.unpack_from("<H", d, pos + 16)[0] + 1
        out.append((pos, n))
        pos += n
    return out


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    tmp = os.environ.get("TMPDIR", "/tmp")
    r1 = gzip.open(os.path.join(SAM, "pe_1.fq.gz")).read()[:400000]
    r2 = gzip.open(os.path.join(SAM, "pe_2.fq.gz")).read()[:400000]
    f1, f2, out = (os.path.join(tmp, "fuzz_bgzf_%d_%s" % (os.getpid(), n)) for n in ("1.fq.gz", "2.fq.gz", "o.sam"))
    open(f2, "wb").write(bgzf(r2, 8000))
    base = bgzf(r1, 8000)
    rng = random.Random(9)
    bad = 0
    for it in range(cases):
        d = bytearray(base)
        kind = it % 4
        if kind == 0:
            for _ in range(rng.randint(1, 4)):
                d[rng.randrange(len(d))] = rng.randrange(256)
        elif kind == 1:
            s, _ = rng.choice(members(d))
            d[s + rng.choice([3, 10, 11, 12, 13, 14, 15, 16, 17])] = rng.randrange(256)
        elif kind == 2:
            d = d[:rng.randrange(20, len(d))]
        else:
            s, n = rng.choice(members(d))
            d[s + n - rng.choice([1, 2, 3, 4, 5, 6, 7, 8])] = rng.randrange(256)
        open(f1, "wb").write(bytes(d))
        r = subprocess.run([BIN, "-silent", "-i", SMALL, "-f", f1, "-f2", f2, "-t", "4", "-o", out], stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
        text = r.stdout.decode(errors="replace")
        if "ERROR: AddressSanitizer" in text or "runtime error" in text or r.returncode not in (0, 1):
            bad += 1
            print("case", it, "kind", kind, "status", r.returncode, text[-600:])
    for f in (f1, f2, out):
        if os.path.exists(f):
            os.remove(f)
    print("%d cases, %d bad" % (cases, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

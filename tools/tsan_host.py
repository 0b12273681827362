#!/usr/bin/env python3
"""ThreadSanitizer over the host pipeline ("race detection"; the reference's own races: src/Mapping.cpp:211-212).

The CPU build of the host pipeline (tests/cpu_backend: mapper.cpp + cli.cpp bound to the CPU oracle -- sanitizers run on the CPU build
only) compiled with -fsanitize=thread, run over the golden inputs and a larger seeded set: reader threads, the worker pool with its
batch hand-over, the EstDistance replay, the writer (shared mappings and pwrite threads), -m, -pacbio with the report of batch k beside
the seeding of batch k + 1, and 2 .. 3 shard processes with the rendezvous block and -parts.  Every run's SAM must equal the plain
binary's, and TSan must print no report.  What it cannot see: the device stream's lanes and seeding groups (stream.inc) exist only
above the HIP backend; tools/stress_groups.py is their stress run on the GPU box.

    python tools/tsan_host.py [--pairs 40000] [--log profiles/r06_tsan.log]
Exit status 0: no report, all outputs identical."""
import argparse
import gzip
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")
PREFIX = os.path.join(GOLD, "idx", "small")
PLAIN = os.path.join(ROOT, "tests", "_build", "kart-host-oracle")
TSAN = os.path.join(ROOT, "tests", "_build", "kart-host-oracle-tsan")


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "tests", "cpu_backend"), PLAIN, "tsan"], stdout=subprocess.DEVNULL)


def runs(tmp, pairs):
    from kart_amd import synth
    from kart_amd.index_build import read_fasta
    genome = {n: s for n, _, s in read_fasta(os.path.join(GOLD, "small.fa"))}
    names, r1, r2 = synth.simulate_pairs(genome, pairs, seed=321, err=0.02, mut=0.003, indel_frac=0.3)
    f1, f2 = os.path.join(tmp, "t_1.fq"), os.path.join(tmp, "t_2.fq")
    synth.write_fastq(f1, names, r1, mate=1)
    synth.write_fastq(f2, names, r2, mate=2)
    gz = {}
    for name in ("pe_1.fq", "pe_2.fq", "pacbio.fq", "se.fq"):
        gz[name] = os.path.join(tmp, name)
        with gzip.open(os.path.join(GOLD, "sam", name + ".gz")) as fi, open(gz[name], "wb") as fo:
            fo.write(fi.read())
    pe = ["-f", f1, "-f2", f2]
    out = []
    for t in (2, 4, 8):
        out.append(("pe -t %d" % t, pe + ["-t", str(t)], {}))
    out.append(("pe -m -t 6", pe + ["-m", "-t", "6"], {}))
    out.append(("pe small batches -t 8", pe + ["-t", "8"], {"KART_AMD_BATCH_READS": "8000"}))
    out.append(("pe pwrite writer -t 6", pe + ["-t", "6"], {"KART_AMD_NO_MMAP_OUT": "1", "KART_AMD_PWRITE_THREADS": "3"}))
    out.append(("pe 3 reader threads -t 6", pe + ["-t", "6"], {"KART_AMD_READER_THREADS": "3"}))
    out.append(("se golden -t 4", ["-f", gz["se.fq"], "-t", "4"], {}))
    # the seeded set as two ordinary gzip files: the several-thread reader (both files at once, chunks of 64 KB: several rounds) in front of the gz reader
    for f in (f1, f2):
        with open(f, "rb") as fi, open(f + ".gz", "wb") as fo:
            fo.write(gzip.compress(fi.read()))
    out.append(("pe gz, several-thread reader -t 8", ["-f", f1 + ".gz", "-f2", f2 + ".gz", "-t", "8"], {"KART_AMD_PGZ_MIN_KB": "0", "KART_AMD_PGZ_CHUNK_KB": "64"}))
    for env in ({}, {"KART_AMD_PACBIO_CHUNKS": "4", "KART_AMD_FRAG_DEPTH": "2"}, {"KART_AMD_LONG_NO_OVERLAP": "1"}):
        out.append(("pacbio golden -t 4 %s" % (env or ""), ["-f", gz["pacbio.fq"], "-pacbio", "-t", "4"], env))
    for dev in ("0,1", "0,1,2"):
        out.append(("pe -gpu %s -t 6" % dev, pe + ["-gpu", dev, "-t", "6"], {}))
        out.append(("pe -gpu %s -parts -t 6" % dev, pe + ["-gpu", dev, "-parts", "-t", "6"], {}))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=40000)
    ap.add_argument("--log", default=None)
    ap.add_argument("--only", type=int, default=0, help="the first N runs (the test suite's short form)")
    args = ap.parse_args()
    build()
    lines, bad = [], 0
    with tempfile.TemporaryDirectory() as tmp:
        todo = runs(tmp, args.pairs)
        if args.only:
            todo = todo[:args.only]
        for name, argv, env in todo:
            outs = {}
            rep = 0
            secs = {}
            for tag, binary in (("plain", PLAIN), ("tsan", TSAN)):
                o = os.path.join(tmp, "o_%s.sam" % tag)
                for f in os.listdir(tmp):
                    if f.startswith("o_%s.sam" % tag):
                        os.remove(os.path.join(tmp, f))
                e = dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=0 report_signal_unsafe=0", **env)
                t0 = time.time()
                r = subprocess.run([binary, "-silent", "-i", PREFIX] + argv + ["-o", o], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=e)
                secs[tag] = time.time() - t0
                parts = sorted(f for f in os.listdir(tmp) if f.startswith("o_%s.sam" % tag))
                outs[tag] = (r.returncode, b"".join(open(os.path.join(tmp, f), "rb").read() for f in parts))
                if tag == "tsan":
                    txt = r.stdout.decode(errors="replace")
                    rep = txt.count("WARNING: ThreadSanitizer")
                    if rep:
                        lines.append(txt[:6000])
            same = outs["plain"] == outs["tsan"] and outs["plain"][0] == 0
            bad += (0 if same else 1) + rep
            lines.append("%-60s plain %5.1f s  tsan %6.1f s  reports %d  output %s" % (name, secs["plain"], secs["tsan"], rep, "identical" if same else "DIFFERS (status %d / %d)" % (outs["plain"][0], outs["tsan"][0])))
            print(lines[-1], flush=True)
    lines.append("ThreadSanitizer: %d runs, %d reports or differences" % (len(todo), bad))
    print(lines[-1])
    if args.log:
        with open(args.log, "w") as fh:
            fh.write("# tools/tsan_host.py --pairs %d: the CPU build of the host pipeline under -fsanitize=thread (g++ %s)\n" % (args.pairs, subprocess.check_output(["g++", "-dumpversion"]).decode().strip()))
            fh.write("\n".join(lines) + "\n")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

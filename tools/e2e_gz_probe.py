#!/usr/bin/env python3
"""kart-amd on gzipped paired FASTQ (E. coli-sized set): how far the serial gz readers hold the pipeline back (MEASUREMENT TOOL, GPU box)."""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, ".")
from kart_amd import synth, index_build
n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 2000000
d = tempfile.mkdtemp(prefix="kart_gz")
genome = synth.make_genome([("decoy", 2000), ("chrE", 4639675)], seed=2, gc=0.508)
fa = os.path.join(d, "g.fa"); synth.write_fasta(fa, genome)
index_build.build_index(fa, os.path.join(d, "idx"))
names, r1, r2 = synth.simulate_pairs(genome, n_pairs, seed=5, err=0.01)
f1, f2 = os.path.join(d, "r1.fq"), os.path.join(d, "r2.fq")
synth.write_fastq(f1, names, r1, mate=1); synth.write_fastq(f2, names, r2, mate=2)
subprocess.run(["gzip", "-1", "-k", f1, f2], check=True)
for tag, a, b in (("plain", f1, f2), ("gz", f1 + ".gz", f2 + ".gz")):
    for exe, t in (("kart_amd/bin/kart-amd", "16"), ("oracle/_ref/kart", "16")):
        if not os.path.exists(exe): continue
        t0 = time.time()
        r = subprocess.run([exe, "-silent", "-i", os.path.join(d, "idx"), "-f", a, "-f2", b, "-t", t, "-o", os.path.join(d, "o.sam")], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                           env=dict(os.environ, KART_AMD_VERBOSE="1"))
        dt = time.time() - t0
        extra = [l.strip() for l in r.stdout.decode().splitlines() if l.startswith(("stage seconds", "mapping seconds"))]
        print(tag, os.path.basename(exe), "wall %.2f s = %.2f M reads/s" % (dt, 2 * n_pairs / dt / 1e6), extra)

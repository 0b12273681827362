#!/usr/bin/env python3
"""kart-amd alone on the E. coli-sized set with KART_AMD_VERBOSE, several thread counts and output targets: where the mapping
wall time goes (MEASUREMENT TOOL, GPU box).  usage: python tools/e2e_stage_probe.py [pairs]"""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, ".")
from kart_amd import synth, index_build
n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 4000000
d = tempfile.mkdtemp(prefix="kart_probe")
genome = synth.make_genome([("decoy", 2000), ("chrE", 4639675)], seed=2, gc=0.508)
fa = os.path.join(d, "g.fa"); synth.write_fasta(fa, genome)
index_build.build_index(fa, os.path.join(d, "idx"))
names, r1, r2 = synth.simulate_pairs(genome, n_pairs, seed=5, err=0.01)
f1, f2 = os.path.join(d, "r1.fq"), os.path.join(d, "r2.fq")
synth.write_fastq(f1, names, r1, mate=1); synth.write_fastq(f2, names, r2, mate=2)
for t, out in ((16, os.path.join(d, "a.sam")), (16, "/dev/null"), (32, os.path.join(d, "a.sam")), (32, "/dev/null"), (24, os.path.join(d, "a.sam"))):
    t0 = time.time()
    r = subprocess.run(["kart_amd/bin/kart-amd", "-silent", "-i", os.path.join(d, "idx"), "-f", f1, "-f2", f2, "-t", str(t), "-o", out], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                       env=dict(os.environ, KART_AMD_VERBOSE="1"))
    print("t=%d out=%s wall %.2f" % (t, "file" if out != "/dev/null" else "null", time.time() - t0))
    for line in r.stdout.decode().splitlines():
        if line.startswith(("stage seconds", "worker thread", "mapping seconds")): print("   ", line.strip())

#!/usr/bin/env python3
"""End-to-end (FASTQ -> SAM) comparison on the GPU box: kart-amd vs the unmodified reference binary
(oracle/_ref/kart) at -t 1 and -t <cores>, same input, byte-compare against -t 1.  TEST/MEASUREMENT TOOL."""
import os, subprocess, sys, time, tempfile, json
sys.path.insert(0, ".")
import numpy as np
from kart_amd import synth, index_build
n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 500000
glen = int(sys.argv[2]) if len(sys.argv) > 2 else 4639675
skip_t1 = len(sys.argv) > 3 and sys.argv[3] == "skip_t1"
d = tempfile.mkdtemp(prefix="kart_e2e")
genome = synth.make_genome([("decoy", 2000), ("chrE", glen)], seed=2, gc=0.508)
fa = os.path.join(d, "g.fa"); synth.write_fasta(fa, genome)
t = time.time(); index_build.build_index(fa, os.path.join(d, "idx")); t_idx = time.time() - t
names, r1, r2 = synth.simulate_pairs(genome, n_pairs, seed=5, err=0.01)
f1, f2 = os.path.join(d, "r1.fq"), os.path.join(d, "r2.fq")
synth.write_fastq(f1, names, r1, mate=1); synth.write_fastq(f2, names, r2, mate=2)
res = {"pairs": n_pairs, "genome": glen, "index_build_s": round(t_idx, 1)}
def run(tag, cmd):
    t = time.time(); r = subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=dict(os.environ, KART_AMD_VERBOSE="1")); dt = time.time() - t
    res[tag] = {"seconds": round(dt, 2), "reads_per_s": round(2 * n_pairs / dt)}
    for line in r.stdout.decode().splitlines():
        if line.startswith("mapping seconds"):
            ms = float(line.split(":")[1])
            res[tag]["mapping_seconds"] = ms
            res[tag]["mapping_reads_per_s"] = round(2 * n_pairs / ms)
        if line.startswith("stage seconds") or line.startswith("worker thread-seconds"):
            res[tag].setdefault("log", []).append(line.strip())
common = ["-silent", "-i", os.path.join(d, "idx"), "-f", f1, "-f2", f2]
for t_ in ((32, 64) if skip_t1 else (8, 32, 64, 128)):
    run("kart_amd_t%d" % t_, ["kart_amd/bin/kart-amd"] + common + ["-t", str(t_), "-o", os.path.join(d, "amd.sam")])
ref = "oracle/_ref/kart"
if os.path.exists(ref):
    if not skip_t1: run("ref_t1", [ref] + common + ["-t", "1", "-o", os.path.join(d, "ref1.sam")])
    cores = os.cpu_count()
    for t_ in sorted({32, 64} if skip_t1 else {8, 32, cores}):
        run("ref_t%d" % t_, [ref] + common + ["-t", str(t_), "-o", os.path.join(d, "refN.sam")])
    if not skip_t1: res["identical_to_ref_t1"] = open(os.path.join(d, "amd.sam"), "rb").read() == open(os.path.join(d, "ref1.sam"), "rb").read()
print(json.dumps(res))

#!/usr/bin/env python3
"""configs[3] in the small: N x 7 kb reads at 15 % error with -pacbio, kart-amd vs the reference binary."""
import os, subprocess, sys, time, tempfile, json
sys.path.insert(0, ".")
from kart_amd import synth, index_build
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
glen = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000_000
d = tempfile.mkdtemp(prefix="kart_pb")
genome = synth.make_genome([("decoy", 2000), ("chrE", glen)], seed=2, gc=0.508)
fa = os.path.join(d, "g.fa"); synth.write_fasta(fa, genome)
index_build.build_index(fa, os.path.join(d, "idx"))
names, reads = synth.simulate_long_reads(genome, n, seed=9, read_len=7000, err=0.15, indel_err_frac=0.1)
fq = os.path.join(d, "long.fq"); synth.write_fastq(fq, names, reads)
res = {"reads": n, "read_len": 7000, "genome": glen}
def run(tag, cmd):
    t = time.time(); r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=dict(os.environ, KART_AMD_VERBOSE="1")); dt = time.time() - t
    res[tag] = {"rc": r.returncode, "seconds": round(dt, 2), "reads_per_s": round(n / dt, 1)}
    for line in r.stdout.decode().splitlines():
        if line.startswith("mapping seconds"): res[tag]["mapping_seconds"] = float(line.split(":")[1])
        if line.startswith("stage seconds") or line.startswith("worker thread-seconds"): res[tag].setdefault("stages", []).append(line)
common = ["-silent", "-i", os.path.join(d, "idx"), "-f", fq, "-pacbio"]
for t_ in (8, 32):
    run("kart_amd_t%d" % t_, ["kart_amd/bin/kart-amd"] + common + ["-t", str(t_), "-o", os.path.join(d, "amd.sam")])
ref = "oracle/_ref/kart"
if os.path.exists(ref):
    run("ref_t32", [ref] + common + ["-t", "32", "-o", os.path.join(d, "refn.sam")])
    if n <= 8000:
        run("ref_t1", [ref] + common + ["-t", "1", "-o", os.path.join(d, "ref1.sam")])
        res["identical_to_ref_t1"] = open(os.path.join(d, "amd.sam"), "rb").read() == open(os.path.join(d, "ref1.sam"), "rb").read()
    else:   # -t 32 prints the same records in a thread-dependent order: compare as multisets of lines
        res["same_records_as_ref_t32"] = sorted(open(os.path.join(d, "amd.sam"), "rb").read().split(b"\n")) == sorted(open(os.path.join(d, "refn.sam"), "rb").read().split(b"\n"))
print(json.dumps(res))

import os, subprocess, sys, time, tempfile
sys.path.insert(0, ".")
from kart_amd import synth, index_build
n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
d = tempfile.mkdtemp(prefix="kart_e2e")
genome = synth.make_genome([("decoy", 2000), ("chrE", 4639675)], seed=2, gc=0.508)
fa = os.path.join(d, "g.fa"); synth.write_fasta(fa, genome)
index_build.build_index(fa, os.path.join(d, "idx"))
names, r1, r2 = synth.simulate_pairs(genome, n_pairs, seed=5, err=0.01)
f1, f2 = os.path.join(d, "r1.fq"), os.path.join(d, "r2.fq")
synth.write_fastq(f1, names, r1, mate=1); synth.write_fastq(f2, names, r2, mate=2)
for t in sys.argv[2:] or ["1", "32"]:
    t0 = time.time()
    r = subprocess.run(["kart_amd/bin/kart-amd", "-silent", "-i", os.path.join(d, "idx"), "-f", f1, "-f2", f2, "-o", os.path.join(d, "o.sam"), "-t", t],
                       env=dict(os.environ, KART_AMD_VERBOSE="1"), stdout=subprocess.PIPE)
    print("threads", t, "wall %.2f" % (time.time() - t0), [l for l in r.stdout.decode().splitlines() if "stage" in l or "re-mapped" in l])

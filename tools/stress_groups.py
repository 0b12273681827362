#!/usr/bin/env python3
"""Repeated kart-amd runs of one paired-end input with the seeding groups on, in several configurations, against the run with the
groups off: every run must write the same bytes.  Prints the first differing lines of a run that does not.  VALIDATION TOOL (GPU box).
usage: python tools/stress_groups.py [runs per configuration]"""
import hashlib, os, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kart_amd import synth
from kart_amd.index_build import read_fasta
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
genome = {n: s for n, _, s in read_fasta(os.path.join(ROOT, "tests", "golden", "small.fa"))}
tmp = tempfile.mkdtemp()
names, r1, r2 = synth.simulate_pairs(genome, 30000, seed=77, err=0.02, mut=0.003, indel_frac=0.3, n_frac=0.0005)
f1, f2 = os.path.join(tmp, "a_1.fq"), os.path.join(tmp, "a_2.fq")
synth.write_fastq(f1, names, r1, mate=1); synth.write_fastq(f2, names, r2, mate=2)


def run(env, tag):
    out = os.path.join(tmp, tag + ".sam")
    r = subprocess.run([os.environ.get("KART_BIN", os.path.join(ROOT, "kart_amd", "bin", "kart-amd")), "-silent", "-i", os.path.join(ROOT, "tests", "golden", "idx", "small"), "-f", f1, "-f2", f2, "-o", out, "-t", "8"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=dict(os.environ, KART_AMD_VERBOSE="1", **env))
    assert r.returncode == 0, r.stdout.decode()[-500:]
    return open(out, "rb").read(), r.stdout.decode()


want, _ = run({"KART_AMD_SEED_GROUP": "0"}, "base")
# (round 4: 224 runs of this matrix clean once the two null-stream memsets of the alignment control block / the workspace's control
#  block were moved onto the workspace's stream; before, 4-11 of 16 runs of the grouped configurations lost the mate rescue of a few pairs)
configs = [("8 lanes g4 turns (default)", {}), ("8 lanes g4 no turns", {"KG_GROUP_NO_TURNS": "1"}), ("4 lanes g4 turns", {"KART_AMD_STREAM_LANES": "4"}),
           ("8 lanes independent", {"KART_AMD_SEED_GROUP": "0", "KART_AMD_STREAM_LANES": "8"}), ("8 lanes g2", {"KART_AMD_SEED_GROUP": "2", "KART_AMD_STREAM_LANES": "8"}),
           ("8 lanes g4 turns 4 k batches", {"KART_AMD_STREAM_READS": "4000"}), ("8 lanes independent 4 k batches", {"KART_AMD_SEED_GROUP": "0", "KART_AMD_STREAM_LANES": "8", "KART_AMD_STREAM_READS": "4000"})]
if len(sys.argv) > 2 and "=" in sys.argv[2]:          # explicit environments: "K=V,K=V" ...
    configs = [(spec, dict(kv.split("=", 1) for kv in spec.split(","))) for spec in sys.argv[2:]]
elif len(sys.argv) > 2:
    configs = [c for c in configs if any(k in c[0] for k in sys.argv[2:])]
for name, env in configs:
    bad = 0
    for i in range(runs):
        got, log = run(env, "g")
        if got != want:
            bad += 1
            if bad == 1:
                a, b = want.split(b"\n"), got.split(b"\n")
                print("  [%s] run %d differs: %d vs %d lines" % (name, i, len(a), len(b)))
                k = 0
                for j, (x, y) in enumerate(zip(a, b)):
                    if x != y:
                        print("   ", j, x[:150].decode()); print("   ", j, y[:150].decode()); k += 1
                        if k >= 4: break
                diff = [j for j, (x, y) in enumerate(zip(a, b)) if x != y]
                print("    differing lines (header = 6 lines, 4000 reads per chunk):", diff[:60], "chunks", sorted(set((j - 6) // 4000 for j in diff)))
                print("   ", [l for l in log.splitlines() if "re-mapped" in l or "device report" in l][-2:])
    print("%-18s %d of %d runs differ" % (name, bad, runs))

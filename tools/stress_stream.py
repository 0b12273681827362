#!/usr/bin/env python3
"""Repeated kart-amd runs of one paired-end input under varying batch sizes / lane counts / thread counts: every run must write the same
bytes (the in-order commit makes the output independent of timing).  VALIDATION TOOL (GPU box).  usage: python tools/stress_stream.py [-m] [runs]"""
import hashlib, os, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kart_amd import synth
from kart_amd.index_build import read_fasta
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
flags = ["-m"] if "-m" in sys.argv else []
runs = int([a for a in sys.argv[1:] if a.isdigit()][0]) if [a for a in sys.argv[1:] if a.isdigit()] else 24
genome = {n: s for n, _, s in read_fasta(os.path.join(ROOT, "tests", "golden", "small.fa"))}
tmp = tempfile.mkdtemp()
f1, f2 = os.path.join(tmp, "a_1.fq"), os.path.join(tmp, "a_2.fq")
with open(f1, "wb") as o1, open(f2, "wb") as o2:
    for part, ins in enumerate((300, 260, 220, 180)):          # a drifting insert size keeps EstDistance moving: many re-mapped chunks
        names, r1, r2 = synth.simulate_pairs(genome, 20000, seed=950 + part, err=0.02, mut=0.003, indel_frac=0.3, ins_mean=float(ins), ins_sd=ins / 8.0)
        names = ["p%d_%s" % (part, n) for n in names]
        p1, p2 = os.path.join(tmp, "t1.fq"), os.path.join(tmp, "t2.fq")
        synth.write_fastq(p1, names, r1, mate=1); synth.write_fastq(p2, names, r2, mate=2)
        o1.write(open(p1, "rb").read()); o2.write(open(p2, "rb").read())
seen = {}
for i in range(runs):
    env = dict(os.environ, KART_AMD_STREAM_READS=str((4000, 8000, 16000, 1000000)[i % 4]), KART_AMD_STREAM_LANES=str(1 + i % 4), KART_AMD_UNSET_FLAG=str(1 << 20))
    if i % 6 == 5:
        env["KART_AMD_NO_STREAM"] = "1"
    out = os.path.join(tmp, "o%d.sam" % i)
    r = subprocess.run([os.path.join(ROOT, "kart_amd", "bin", "kart-amd"), "-silent", "-i", os.path.join(ROOT, "tests", "golden", "idx", "small"), "-f", f1, "-f2", f2, "-o", out, "-t", str(2 + 3 * (i % 5))] + flags,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=dict(env, KART_AMD_VERBOSE="1"))
    assert r.returncode == 0, r.stdout.decode()[-500:]
    h = hashlib.sha1(open(out, "rb").read()).hexdigest()
    remap = [l for l in r.stdout.decode().splitlines() if l.startswith("chunks re-mapped")]
    seen.setdefault(h, []).append((i, env.get("KART_AMD_STREAM_READS"), env.get("KART_AMD_STREAM_LANES"), "nostream" if "KART_AMD_NO_STREAM" in env else "stream", remap[-1] if remap else ""))
    if len(seen) > 1:
        break
for h, v in seen.items():
    print(h, len(v), v[:3])
if len(seen) > 1:
    a, b = [open(os.path.join(tmp, "o%d.sam" % v[0][0]), "rb").read().split(b"\n") for v in seen.values()][:2]
    print("lines", len(a), len(b))
    n = 0
    for i, (x, y) in enumerate(zip(a, b)):
        if x != y:
            print(i, x[:140]); print(i, y[:140]); n += 1
            if n > 6: break
    sys.exit(1)
print("all %d runs identical" % runs)

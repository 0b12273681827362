#!/usr/bin/env python3
"""Device report vs host report, record by record (GPU box).  Runs kart-amd with KART_AMD_CHECK_ALIGN=1: every read is mapped by
the host implementation as well, and the SAM text made from the device's kg_aln_record is compared with the host's text.
DEBUG / VALIDATION TOOL.  usage: python tools/check_align.py [pairs]"""
import gzip, os, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_host_pipeline import CASES, materialise, SAM, GOLD_OF
EXE = os.path.join(ROOT, "kart_amd", "bin", "kart-amd")
PREFIX = os.path.join(ROOT, "tests", "golden", "idx", "small")
tmp = tempfile.mkdtemp()
bad = 0


def run(tag, args, want=None, env=None):
    global bad
    out = os.path.join(tmp, tag + ".sam")
    e = dict(os.environ, KART_AMD_CHECK_ALIGN="1")
    e.update(env or {})
    r = subprocess.run([EXE, "-silent", "-i", PREFIX] + args + ["-o", out, "-t", "8"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e)
    summ = [l for l in r.stdout.decode().splitlines() if l.startswith("CHECK_ALIGN")]
    same = None if want is None else (open(out, "rb").read() == want)
    print(tag, "rc", r.returncode, summ, "golden identical:" if want is not None else "", same if want is not None else "")
    err = r.stderr.decode()
    if err.strip():
        print(err[:3000])
    if r.returncode != 0 or (summ and not summ[0].endswith(" 0 differ")) or same is False:
        bad += 1
    # the same run with the device records actually used
    r2 = subprocess.run([EXE, "-silent", "-i", PREFIX] + args + ["-o", out + ".dev", "-t", "8"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, KART_AMD_VERBOSE="1"))
    if r2.returncode != 0:
        print("   device-records run FAILED rc", r2.returncode, r2.stderr.decode()[-500:])
        bad += 1
    print("   ", [l for l in r2.stdout.decode().splitlines() if l.startswith("worker thread-seconds") or "re-mapped" in l or l.startswith("device report")])
    if want is not None:
        ok = open(out + ".dev", "rb").read() == want
        print("   device records used: golden identical:", ok)
        if not ok:
            bad += 1
    return out + ".dev"


for case in ("pe", "pe_plain", "pe_g2", "pe_interleaved", "se", "se_fasta", "edge_pe", "edge_se", "edge_multi_lib", "pe_m", "se_m", "edge_se_m"):
    args = [materialise(tmp, a) if a.endswith((".fq", ".fa", ".gz")) else a for a in CASES[case]]
    want = gzip.open(os.path.join(SAM, GOLD_OF.get(case, case) + ".sam.gz")).read()
    # (-m prints records whose FLAG the reference never assigns: the goldens hold the product's default for those, 0)
    run(case, args, want)

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 30000
from kart_amd import synth
from kart_amd.index_build import read_fasta
genome = {n: s for n, _, s in read_fasta(os.path.join(ROOT, "tests", "golden", "small.fa"))}
for seed, err, ins in ((77, 0.02, 500.0), (78, 0.01, 300.0), (79, 0.04, 200.0)):
    names, r1, r2 = synth.simulate_pairs(genome, pairs, seed=seed, err=err, mut=0.003, indel_frac=0.3, n_frac=0.0005, ins_mean=ins, ins_sd=ins / 8)
    f1, f2 = os.path.join(tmp, "a_1.fq"), os.path.join(tmp, "a_2.fq")
    synth.write_fastq(f1, names, r1, mate=1)
    synth.write_fastq(f2, names, r2, mate=2)
    dev = run("live_%d" % seed, ["-f", f1, "-f2", f2])
    ref = os.path.join(ROOT, "oracle", "_ref", "kart")
    if os.path.exists(ref):
        o = os.path.join(tmp, "ref.sam")
        subprocess.run([ref, "-silent", "-i", PREFIX, "-f", f1, "-f2", f2, "-o", o, "-t", "1"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        same = open(o, "rb").read() == open(dev, "rb").read()
        print("   vs live reference -t 1:", same)
        if not same:
            bad += 1
            a, b = open(o, "rb").read().split(b"\n"), open(dev, "rb").read().split(b"\n")
            shown = 0
            for x, y in zip(a, b):
                if x != y and shown < 5:
                    print("     ref:", x[:200].decode()); print("     dev:", y[:200].decode()); shown += 1
    run("live_se_%d" % seed, ["-f", f1])
    # -m: every record the output loops print; never-assigned FLAGs print as 3000 on both sides of the comparison
    run("live_m_%d" % seed, ["-f", f1, "-f2", f2, "-m"], env={"KART_AMD_UNSET_FLAG": "3000"})
    run("live_se_m_%d" % seed, ["-f", f1, "-m"], env={"KART_AMD_UNSET_FLAG": "3000"})
print("FAILED" if bad else "ALL OK", bad)
sys.exit(1 if bad else 0)

#!/bin/bash
# round 6, sixteenth GPU call: the damaged gz stream through the device stream -- the lines that differ from the reference's
mkdir -p gpurun_out
export TMPDIR=/tmp
python - > gpurun_out/r06p_damaged_gz.log 2>&1 <<'PY'
import gzip, os, subprocess
ROOT = os.environ.get("GRAFT_REPO_ROOT", ".")
SAM = os.path.join(ROOT, "tests", "golden", "sam")
IDX = os.path.join(ROOT, "tests", "golden", "small")
IDX = os.path.join(ROOT, "tests", "golden", "idx", "small")
r1 = gzip.open(os.path.join(SAM, "pe_1.fq.gz")).read()
r2 = gzip.open(os.path.join(SAM, "pe_2.fq.gz")).read()
os.makedirs("/tmp/dmg", exist_ok=True)
d = bytearray(gzip.compress(r1)); d[len(d) * 6 // 10] ^= 0x55
open("/tmp/dmg/e1.fq.gz", "wb").write(bytes(d)); open("/tmp/dmg/e2.fq.gz", "wb").write(gzip.compress(r2))
def run(binary, t, env, out):
    r = subprocess.run([binary, "-silent", "-i", IDX, "-f", "/tmp/dmg/e1.fq.gz", "-f2", "/tmp/dmg/e2.fq.gz", "-t", t, "-o", out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=dict(os.environ, **env))
    print(binary.split("/")[-1], env, "rc", r.returncode)
    return open(out, "rb").read().split(b"\n")
ref = run(os.path.join(ROOT, "oracle", "_ref", "kart"), "1", {}, "/tmp/dmg/r.sam")
base = {"KART_AMD_PGZ_MIN_KB": "0", "KART_AMD_PGZ_CHUNK_KB": "16"}
for name, env in (("stream", base), ("host gz reader", dict(base, KART_AMD_NO_GZ_STREAM="1")), ("stream, zlib", dict(base, KART_AMD_NO_PGZ="1"))):
    got = run(os.path.join(ROOT, "kart_amd", "bin", "kart-amd"), "16", env, "/tmp/dmg/o.sam")
    diff = [i for i, (x, y) in enumerate(zip(ref, got)) if x != y]
    print(name, "lines", len(ref), len(got), "differing", diff[:10])
    for i in diff[:4]:
        print("  ref:", ref[i][:400]); print("  got:", got[i][:400])
PY
cat gpurun_out/r06p_damaged_gz.log | cut -c1-600

#!/usr/bin/env python3
"""End-to-end (FASTQ -> SAM) on the hg38-sized synthetic genome of bench.py: kart-amd vs the unmodified reference binary at
-t <threads>, same files.  MEASUREMENT TOOL (GPU box).  usage: python tools/e2e_large.py [genome_len] [pairs] [check_t1_pairs]"""
import json, os, subprocess, sys, time
import torch
sys.path.insert(0, ".")
import bench
dev = torch.device("cuda", 0)
L = int(sys.argv[1]) if len(sys.argv) > 1 else bench.HG38_LEN
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
wd = bench.pick_workdir(60 << 30)
prefix = os.path.join(wd, "synth_v2_%d" % L)
subprocess.run([sys.executable, "bench.py", "--genome-len", str(L), "--pairs", "1000000", "--leg", "seeding", "--seed-steps", "1"], stdout=subprocess.DEVNULL)   # builds + caches the index
codes = bench.make_large_codes(L, 3, dev)
f1, f2 = os.path.join(wd, "l1.fq"), os.path.join(wd, "l2.fq")
bench.write_fastq_pairs(codes, pairs, 5, f1, f2, dev)
del codes; torch.cuda.empty_cache()
threads = int(os.environ.get("E2E_THREADS", min(32, 2 * bench.effective_cores())))
res = {"genome_len": L, "reads": 2 * pairs, "threads": threads}
def run(tag, exe, t):
    t0 = time.time()
    r = subprocess.run([exe, "-silent", "-i", prefix, "-f", f1, "-f2", f2, "-o", os.environ.get("E2E_OUT") or os.path.join(wd, tag + os.environ.get("E2E_OUT_SUFFIX", "") + ".sam"), "-t", str(t)], stdout=subprocess.PIPE,
                       stderr=open(os.path.join("gpurun_out", tag + ".stderr"), "wb") if os.path.isdir("gpurun_out") else subprocess.DEVNULL,
                       env=dict(os.environ, KART_AMD_VERBOSE="1"))
    dt = time.time() - t0
    res[tag] = {"rc": r.returncode, "process_seconds": round(dt, 2), "reads_per_s_process": round(2 * pairs / dt)}
    for line in r.stdout.decode().splitlines():
        if line.startswith("mapping seconds"):
            ms = float(line.split(":")[1]); res[tag]["mapping_seconds"] = ms; res[tag]["reads_per_s_mapping_phase"] = round(2 * pairs / ms)
        if line.startswith(("stage seconds", "worker thread-seconds", "All the", "device report", "CHECK_ALIGN", "chunks re-mapped", "cpu seconds")):
            res[tag].setdefault("log", []).append(line.strip())
run("kart_amd", "kart_amd/bin/kart-amd", threads)
for i, setting in enumerate(filter(None, os.environ.get("E2E_SWEEP", "").split(";"))):       # e.g. "KART_AMD_WRITER_AHEAD=1;;KART_AMD_WRITER_MODE=0"
    kv = dict(x.split("=", 1) for x in setting.split(",") if "=" in x)
    os.environ.update(kv)
    run("sweep%d" % i, "kart_amd/bin/kart-amd", threads)
    res["sweep%d" % i]["setting"] = setting
    for k in kv:
        del os.environ[k]
if os.environ.get("E2E_CHECK_ALIGN"):      # every device record against the host's text for the same read (slow: the host maps everything too)
    os.environ["KART_AMD_CHECK_ALIGN"] = "1"
    run("kart_amd_check_align", "kart_amd/bin/kart-amd", threads)
    del os.environ["KART_AMD_CHECK_ALIGN"]
if os.path.exists("oracle/_ref/kart") and not os.environ.get("E2E_NO_REF"):
    run("reference_kart", "oracle/_ref/kart", threads)
    # the reference prints its own mapping time ("... processed in N seconds"), which excludes its index load
print(json.dumps(res))

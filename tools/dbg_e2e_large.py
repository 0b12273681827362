import os, sys, subprocess, torch
sys.path.insert(0, ".")
import bench
dev = torch.device("cuda", 0)
L = int(sys.argv[1]) if len(sys.argv) > 1 else bench.HG38_LEN
wd = "/tmp/kart_bench_%d" % os.getuid()
prefix = os.path.join(wd, "synth_v2_%d" % L)
subprocess.run([sys.executable, "bench.py", "--genome-len", str(L), "--pairs", "1000000", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-e2e"], stdout=subprocess.DEVNULL)
codes = bench.make_large_codes(L, 3, dev)
f1, f2 = os.path.join(wd, "d1.fq"), os.path.join(wd, "d2.fq")
bench.write_fastq_from_codes(codes, int(sys.argv[2]) if len(sys.argv) > 2 else 100000, 5, f1, f2, dev)
del codes; torch.cuda.empty_cache()
r = subprocess.run(["kart_amd/bin/kart-amd", "-i", prefix, "-f", f1, "-f2", f2, "-o", os.path.join(wd, "d.sam"), "-t", "32"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=dict(os.environ, KART_AMD_VERBOSE="1"))
print("rc", r.returncode); print(r.stdout.decode()[-int(os.environ.get("TAIL", "1500")):])

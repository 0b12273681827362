#!/usr/bin/env python3
"""Bit-exact parity of the GPU seeding path against the CPU oracle on the hg38-sized synthetic index, N reads
(the bench's own spot check covers 4000).  VALIDATION TOOL (GPU box).  usage: python tools/parity_large.py [reads]"""
import os, subprocess, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from kart_amd import api
from oracle import oracle as O
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
L = bench.HG38_LEN
dev = torch.device("cuda", 0)
wd = "/tmp/kart_bench_%d" % os.getuid()
prefix = os.path.join(wd, "synth_v2_%d" % L)
subprocess.run([sys.executable, "bench.py", "--pairs", "1000000", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-e2e"], stdout=subprocess.DEVNULL)
codes = bench.make_large_codes(L, 3, dev)
enc, off = bench.gen_reads_device(codes, n_reads // 2, seed=4242, err=0.011, dev=dev)
del codes
ix = api.Index(prefix, 0, api.KG_SA_FULL)
enc_h, off_h = enc.cpu().numpy(), off.cpu().numpy()
ws = ix.workspace(n_reads, len(enc_h))
t = time.time(); so_g, s_g = ws.seed_batch(enc_h, off_h, 0); tg = time.time() - t
orc = O.Oracle(prefix)
t = time.time(); so_o, s_o = orc.seed_batch(enc_h, off_h, 0, threads=bench.effective_cores()); to = time.time() - t
same = bool((so_g == so_o).all() and (s_g == s_o.astype(api.SEED_DT)).all())
print({"reads": n_reads, "seeds": int(so_o[-1]), "identical": same, "gpu_host_form_s": round(tg, 2), "oracle_s": round(to, 1)})
sys.exit(0 if same else 1)

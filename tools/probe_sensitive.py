#!/usr/bin/env python3
"""SensitiveMode seeding alone (GPU box): kernel times and work counters of kg_seed_batch_device on N x 7 kb reads at 15 % error on
the hg38-sized index of bench.py, at several batch sizes.  MEASUREMENT TOOL.  usage: python tools/probe_sensitive.py [reads ...]"""
import os, subprocess, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from kart_amd import api

sizes = [int(x) for x in sys.argv[1:]] or [10_000, 40_000, 160_000]
L = bench.HG38_LEN
dev = torch.device("cuda", 0)
wd = bench.pick_workdir(80 << 30)
prefix = os.path.join(wd, "synth_v2_%d" % L)
subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--pairs", "1000000", "--leg", "seeding", "--seed-steps", "1"], stdout=subprocess.DEVNULL)
codes = bench.make_large_codes(L, 3, dev)
ix = api.Index(prefix, 0, api.KG_SA_FULL)
RL = 7000
for n in sizes:
    g = torch.Generator(device=dev); g.manual_seed(5)
    enc = torch.empty(n * RL, dtype=torch.uint8, device=dev)
    ar = torch.arange(RL, device=dev)
    for s in range(0, n, 20000):
        m = min(20000, n - s)
        pos = bench.DECOY_LEN + (torch.rand(m, generator=g, device=dev, dtype=torch.float64) * (L - RL - 1)).long()
        r = codes[pos[:, None] + ar]
        e = torch.rand(r.shape, generator=g, device=dev) < 0.15
        r = torch.where(e, (r + torch.randint(1, 4, r.shape, generator=g, device=dev, dtype=torch.uint8)) & 3, r)
        enc[s * RL:(s + m) * RL] = r.reshape(-1)
    off = torch.arange(n + 1, dtype=torch.int64, device=dev) * RL
    cap = 1200 * n + 1024
    d_off = torch.empty(n + 1, dtype=torch.int64, device=dev)
    d_seeds = torch.empty(cap * 16, dtype=torch.uint8, device=dev)
    ws = api.Workspace(ix, n, n * RL)
    ws.set_profiling(True)
    st = torch.cuda.current_stream(dev).cuda_stream
    for it in range(2):
        ws.seed_batch_device(enc.data_ptr(), off.data_ptr(), n, n * RL, d_off.data_ptr(), d_seeds.data_ptr(), cap, api.KG_MODE_SENSITIVE, stream=st)
        torch.cuda.synchronize(dev)
    assert ws.overflow() == 0
    km = ws.kernel_ms()
    c = ws.counters().as_dict(); t = ws.traffic().as_dict()
    print(json.dumps({"reads": n, "kernels_ms": [round(float(x), 2) for x in km], "seeds_per_read": int(d_off[n]) / n,
                      "per_read": {k: round(v / n, 1) for k, v in c.items()}, "fetched_per_read": {k: round(v / n, 1) for k, v in t.items()}}))
    del ws, d_seeds, enc

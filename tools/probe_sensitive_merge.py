#!/usr/bin/env python3
"""Do SensitiveMode chains from different start positions merge (GPU box)?  The chain of IdentifySeedPairs_SensitiveMode
(src/AlignmentCandidates.cpp:132-169) has the state `pos` alone; a read seeded again from position x (= the read without its first x
bases) follows the same chain as the whole read from the first position both chains visit.  For N x 7 kb reads at 15 % error on the
hg38-sized index: where does the chain of read[x:] first share a hit start with the chain of read[0:] -- the distance a speculative
segment start needs before its results are the sequential ones.  MEASUREMENT TOOL for DESIGN 8-1."""
import os, subprocess, sys, json
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from kart_amd import api

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
L = bench.HG38_LEN
dev = torch.device("cuda", 0)
wd = bench.pick_workdir(80 << 30)
prefix = os.path.join(wd, "synth_v2_%d" % L)
subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--pairs", "1000000", "--leg", "seeding", "--seed-steps", "1"], stdout=subprocess.DEVNULL)
codes = bench.make_large_codes(L, 3, dev)
ix = api.Index(prefix, 0, api.KG_SA_FULL40)
RL = 7000
g = torch.Generator(device=dev); g.manual_seed(5)
ar = torch.arange(RL, device=dev)
pos = bench.DECOY_LEN + (torch.rand(n, generator=g, device=dev, dtype=torch.float64) * (L - RL - 1)).long()
r = codes[pos[:, None] + ar]
for err in (0.15, 0.05):
    e = torch.rand(r.shape, generator=g, device=dev) < err
    reads = torch.where(e, (r + torch.randint(1, 4, r.shape, generator=g, device=dev, dtype=torch.uint8)) & 3, r).cpu().numpy()
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)

    def hit_starts(shift):
        """per read: sorted array of the read positions (in whole-read coordinates) where a hit of the chain from `shift` starts"""
        enc = acgt[reads[:, shift:]].reshape(-1)
        off = np.arange(n + 1, dtype=np.int64) * (RL - shift)
        ws = api.Workspace(ix, n, len(enc))
        so, seeds = ws.seed_batch(enc, off, api.KG_MODE_SENSITIVE | api.KG_INPUT_ASCII)
        out = []
        for i in range(n):
            out.append(np.unique(seeds["rPos"][so[i]:so[i + 1]]) + shift)
        ws.close()
        return out

    base = hit_starts(0)
    res = {}
    for shift in (512, 517, 1031, 2048):
        other = hit_starts(shift)
        dist = []
        for a, b in zip(base, other):
            a = a[a >= shift]
            common = np.intersect1d(a, b)
            dist.append(int(common[0] - shift) if len(common) else -1)
        d = np.array(dist)
        ok = d[d >= 0]
        res[shift] = {"never": int((d < 0).sum()), "median": float(np.median(ok)), "p90": float(np.percentile(ok, 90)), "p99": float(np.percentile(ok, 99)),
                      "within_128": float((ok <= 128).mean()), "within_256": float((ok <= 256).mean()), "within_512": float((ok <= 512).mean())}
    print(json.dumps({"error_rate": err, "reads": n, "hits_per_read": float(np.mean([len(x) for x in base])), "first_common_hit_after_start": res}))

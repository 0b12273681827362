#!/usr/bin/env python3
"""NW kernel throughput (GCUPS = DP cells per second) on device-resident batches, per size class."""
import os, sys, time, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kart_amd import api
ix = api.Index(os.path.join(ROOT, "tests", "golden", "idx", "small"), 0, api.KG_SA_SAMPLED)
L = ix.lib
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
res = {}
for name, n, lo, hi in (("tiny_1-8", 8_000_000, 1, 9), ("small_9-32", 2_000_000, 9, 33), ("mid_33-128", 200_000, 33, 129), ("long_300-1100", 4_000, 300, 1101)):
    m = rng.integers(lo, hi, size=n); k = np.clip(m + rng.integers(-3, 4, size=n), lo, hi - 1)
    o1 = np.zeros(n + 1, np.int64); o2 = np.zeros(n + 1, np.int64)
    np.cumsum(m, out=o1[1:]); np.cumsum(k, out=o2[1:])
    f1 = torch.randint(0, 4, (int(o1[-1]) + 16,), device=dev, dtype=torch.uint8)
    f2 = torch.randint(0, 4, (int(o2[-1]) + 16,), device=dev, dtype=torch.uint8)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    f1 = lut[f1.long()]; f2 = lut[f2.long()]
    d1 = torch.from_numpy(o1).to(dev); d2 = torch.from_numpy(o2).to(dev)
    ops = torch.empty(int(o1[-1] + o2[-1]) + 16, dtype=torch.uint8, device=dev)
    ln = torch.empty(n, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    max_len = int(max(m.max(), k.max()))      # (outside the timed calls: a numpy reduction over millions of lengths costs more than the kernels)
    def call():
        rc = L.kg_nw_batch_device(ix.h, f1.data_ptr(), d1.data_ptr(), f2.data_ptr(), d2.data_ptr(), n, max_len, ops.data_ptr(), ln.data_ptr(), stream)
        assert rc == 0, L.kg_last_error()
    call(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3): call()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 3
    cells = float((m.astype(np.float64) * k).sum())
    res[name] = {"pairs": n, "ms": round(dt * 1e3, 3), "pairs_per_s": round(n / dt), "GCUPS": round(cells / dt / 1e9, 2)}
print(json.dumps(res))

#!/usr/bin/env python3
"""What the device and its copy engines did with the time of a mapping run: from a rocprofv3 --kernel-trace --memory-copy-trace pass (csv), inside the window
from the first FASTQ kernel to the last SAM kernel: how long kernels ran (union over streams), how long copies to the host / to the device ran, how
much of that overlapped, and how long the device sat with neither.  DIAGNOSTIC (GPU box).
usage: python tools/timeline_overlap.py <rocprofv3 output dir>"""
import csv, glob, json, sys
import numpy as np


def union(iv):
    """total length of the union of intervals [(a, b)], and the merged list"""
    if not iv:
        return 0, []
    iv = sorted(iv)
    out = [list(iv[0])]
    for a, b in iv[1:]:
        if a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return sum(b - a for a, b in out), out


def intersect(x, y):
    i = j = 0
    tot = 0
    while i < len(x) and j < len(y):
        a, b = max(x[i][0], y[j][0]), min(x[i][1], y[j][1])
        if a < b:
            tot += b - a
        if x[i][1] < y[j][1]:
            i += 1
        else:
            j += 1
    return tot


d = sys.argv[1]
kern, h2d, d2h = [], [], []
per = {}
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "kg::" not in n:
            continue
        a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        kern.append((a, b, n.replace("void ", "").split("(")[0].replace("kg::", "")))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        kind = (r.get("Direction") or r.get("Name") or "").upper()
        (d2h if ("DEVICE_TO_HOST" in kind or "DTOH" in kind) else h2d if ("HOST_TO_DEVICE" in kind or "HTOD" in kind) else []).append((a, b))
fq = [k for k in kern if k[2].startswith("fq_")]
sam = [k for k in kern if k[2].startswith("sam_format")]
if not fq or not sam:
    print(json.dumps({"error": "no stream kernels in the trace", "kernels": len(kern)}))
    sys.exit(0)
w0, w1 = min(k[0] for k in fq), max(k[1] for k in sam)
clip = lambda iv: [(max(a, w0), min(b, w1)) for a, b in iv if b > w0 and a < w1]
K, Km = union(clip([(a, b) for a, b, _ in kern]))
D, Dm = union(clip(d2h))
H, Hm = union(clip(h2d))
C, Cm = union(clip(d2h) + clip(h2d))
busy, _ = union(clip([(a, b) for a, b, _ in kern]) + clip(d2h) + clip(h2d))
for a, b, n in kern:
    if b > w0 and a < w1:
        e = per.setdefault(n, [0, 0])
        e[0] += 1; e[1] += b - a
W = w1 - w0
out = {"window_ms": W / 1e6, "kernels_ms": K / 1e6, "copies_to_host_ms": D / 1e6, "copies_to_device_ms": H / 1e6, "any_copy_ms": C / 1e6,
       "kernels_while_copying_to_host_ms": intersect(Km, Dm) / 1e6, "kernels_while_any_copy_ms": intersect(Km, Cm) / 1e6,
       "neither_kernel_nor_copy_ms": (W - busy) / 1e6,
       "share": {"kernels": K / W, "copies_to_host": D / W, "copies_to_device": H / W, "idle": (W - busy) / W},
       "kernel_ms_summed_per_name": {n: round(t / 1e6, 2) for n, (c, t) in sorted(per.items(), key=lambda x: -x[1][1])[:24]},
       "what": "union over streams inside [first fq_* kernel, last sam_format kernel] of the process (warm-up and timed steps alike)"}
print(json.dumps(out, indent=1))

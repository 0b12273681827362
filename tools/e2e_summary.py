#!/usr/bin/env python3
"""prints the stage lines of a tools/e2e_large.py result read from stdin (helper for sweeps on the GPU box)"""
import json, sys
d = json.load(open(sys.argv[1]) if len(sys.argv) > 1 else sys.stdin)
tag = sys.argv[1] if len(sys.argv) > 1 else ""
for key in sorted(d):
    if key.startswith("sweep"):
        print(key, d[key].get("setting"), "mapping s", d[key].get("mapping_seconds"), [l for l in d[key].get("log", []) if l.startswith("cpu seconds")])
k = d["kart_amd"]
for l in k.get("log", [])[: int(sys.argv[2]) if len(sys.argv) > 2 else 3]:
    print("   ", l[:400])
print(tag, "threads", d["threads"], "mapping s", k.get("mapping_seconds"), "reads/s", k.get("reads_per_s_mapping_phase"))

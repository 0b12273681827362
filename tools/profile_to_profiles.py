#!/usr/bin/env python3
"""Condense the output of tools/profile_round.sh (gpurun_out/<tag>_trace, gpurun_out/<tag>_pmc_*) into the two
files the bench and the docs cite: profiles/<tag>_kernel_stats.csv and profiles/<tag>_pmc_summary.json.
usage: python tools/profile_to_profiles.py <tag> <reads_per_launch> <genome_len>"""
import collections, csv, glob, json, os, shutil, sys

tag, n_reads, glen = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")
stats = sorted(glob.glob(os.path.join(root, "gpurun_out", tag + "_trace", "**", "*kernel_stats.csv"), recursive=True))
if stats:
    shutil.copy(stats[-1], os.path.join(out, tag + "_kernel_stats.csv"))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(os.path.join(root, "gpurun_out", tag + "_pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("void ", "").split("(")[0]
        if "kg::" in name:
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
summ = {k: {c: {"dispatches": len(v), "mean_per_launch": sum(v) / len(v)} for c, v in sorted(cs.items())} for k, cs in sorted(agg.items())}
for k, cs in list(summ.items()):
    if "search_kernel" in k and "TCC_EA0_RDREQ_128B_sum" in cs:
        r128 = cs["TCC_EA0_RDREQ_128B_sum"]["mean_per_launch"]
        r64 = cs.get("TCC_EA0_RDREQ_64B_sum", {"mean_per_launch": 0})["mean_per_launch"]
        wr = cs.get("WRITE_SIZE", {"mean_per_launch": 0})["mean_per_launch"] * 1024          # WRITE_SIZE is in KiB
        fetch = cs.get("FETCH_SIZE", {"mean_per_launch": 0})["mean_per_launch"] * 1024
        rd = r128 * 128 + r64 * 64
        summ["_search_traffic"] = {
            "kernel": k.replace("kg::", ""), "reads_per_launch": n_reads, "genome_len": glen,
            "sa_mode": "compact" if ", 2" in k else "dense" if ", 1" in k else "full",      # (search_kernel<idx_t, kRaw, kSa, kSeg>: kSa 2 = 5-byte entries)
            "FETCH_SIZE_bytes_as_reported": fetch, "WRITE_SIZE_bytes": wr, "RDREQ_128B": r128, "RDREQ_64B": r64,
            "read_bytes_corrected": rd,
            "note": "gfx950: FETCH_SIZE tallies 128-B requests at 64 B (MI355X_MICROARCH.md HBM section); read traffic = RDREQ_128B x 128 + RDREQ_64B x 64.",
            "traffic_bytes_per_launch": rd + wr}
json.dump(summ, open(os.path.join(out, tag + "_pmc_summary.json"), "w"), indent=1)
t = summ.get("_search_traffic")
print("kernel stats:", bool(stats), "| kernels with counters:", len(agg), "| search traffic GB/launch:", round(t["traffic_bytes_per_launch"] / 1e9, 2) if t else None)

#!/usr/bin/env python3
"""usage: python3 tools/kernels_table.py <bench log> -- the `kernels` / `roofline` entries of a bench line as the markdown table of DESIGN.md section 6
(kernel, what it moves per unit, ms per step, achieved GB/s on its algorithmic bytes, fraction of the HBM peak, counter traffic per step)."""
import json, sys

WHAT = {
    "search": "bytes the implemented search needs: 8 B per q-mer entry, 2 x 16 B per single rank step, header + segment per rank of a double / triple step, one SA entry, 48 B per text round, read words, 32 B per hit (~916 B per read)",
    "chain": "32 B per seed (in 16, out 16) + 32 B per candidate + 24 B per read",
    "aln_pair": "44 B per candidate + 16 B per read",
    "aln_rescue": "per rescue window: task 40 B + window text 412 B + the mate's 150 characters + 12 seeds",
    "aln_plan_fast": "per candidate: candidate 32 B + report 78 B + 150 characters + 38 B of text; 16 B per seed",
    "aln_plan": "the candidates left to it: the same + 536 B spill slot; 24 B per NW job",
    "aln_partition": "per task: task 32 B + both fragments (255 B) + plan and pieces 64 B",
    "nw": "integer DP, priced in cells per second (`nw_kernels`): VALU-bound",
    "aln_finish": "per parked candidate: spill slot 536 B + report 78 B + 150 characters; op strings twice",
    "aln_final": "112 B record per read + 90 B per candidate",
    "sam_size": "140 B per read (record + record table in, length out)",
    "sam_format": "0.75 x the FASTQ text (names, bases, qualities) + the SAM text + 144 B per read",
    "fq_parse": "the FASTQ text twice (line count, line index) + 40 B per read",
    "fq_materialise": "308 B per read (bases in, characters out, offsets)",
}
d = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")][-1]
k, r = d["kernels"], d["roofline"]
rows = [("search_kernel", r["search_kernel_ms_per_step"], r["achieved"], r["frac"], (r.get("traffic") or 0) * r["launches"] / d["steps"], "search")]
for n, e in k.items():
    if isinstance(e, dict) and "ms_per_step" in e:
        rows.append((n + ("_kernel" if n != "nw" else " (three kernels)"), e["ms_per_step"], e.get("achieved"), e.get("frac"), e.get("traffic") or 0, n))
rows.sort(key=lambda x: -x[1])
print("| kernel | algorithmic bytes per unit | ms per step | achieved GB/s | frac of 8 TB/s | counter traffic GB per step (traffic / 8 TB/s) |")
print("|---|---|---|---|---|---|")
for name, ms, gbs, frac, tr, key in rows:
    print("| `%s` | %s | %.1f | %s | %s | %s |" % (name, WHAT.get(key, ""), ms, "%.0f" % gbs if gbs else "-", "%.3f" % frac if frac else "-",
                                                   ("%.0f (%.2f)" % (tr / 1e9, tr / (ms * 1e-3) / 8e12)) if tr else "-"))
print("\nsum of the timed kernels: %.0f ms per %.0f ms step; `value` %.1f M mapped reads/s" % (k["timed_kernel_ms_per_step"], d["ms_per_step"], d["value"] / 1e6))

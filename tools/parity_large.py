#!/usr/bin/env python3
"""Bit-exact parity of the GPU seeding path against the CPU oracle on a large synthetic index, N reads (the bench's own
spot check covers a sample of its batch).  VALIDATION TOOL (GPU box); also run by tests/test_large_gpu.py.

    python tools/parity_large.py [--genome-len L] [--reads N] [--sa full|sampled] [--workdir DIR] [--seed S]

Default genome = the hg38-sized one of bench.py (its index is taken from, or built into, bench.py's work directory);
any other length gets its own index in --workdir, built on the GPU by kart_amd.index_build.  KG_FORCE_U64=1 in the
environment selects the 64-bit kernel variants an hg38-sized text uses, whatever the genome size."""
import argparse, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from kart_amd import api, index_build
from oracle import oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--genome-len", type=int, default=bench.HG38_LEN)
ap.add_argument("--reads", type=int, default=2_000_000)
ap.add_argument("--sa", choices=["full", "sampled"], default="full")
ap.add_argument("--workdir", default=None)
ap.add_argument("--seed", type=int, default=4242)
ap.add_argument("--cand-reads", type=int, default=5000, help="reads whose chained candidates are compared as well")
args = ap.parse_args()
n_reads, L = args.reads & ~1, args.genome_len
dev = torch.device("cuda", 0)
wd = args.workdir or bench.pick_workdir(20 << 30)
os.makedirs(wd, exist_ok=True)
prefix = os.path.join(wd, "synth_v2_%d" % L)
codes = bench.make_large_codes(L, 3, dev)
if not all(os.path.exists(prefix + e) for e in (".bwt", ".sa", ".pac", ".ann", ".amb")):
    anns = [("decoy", "(null)", 0, bench.DECOY_LEN, 0)] + [(nm, "(null)", off, ln, 0) for nm, off, ln in bench.contig_table(L)]
    index_build.build_index_from_codes(codes.cpu().numpy(), anns, [], prefix, device=str(dev))
    torch.cuda.empty_cache()
enc, off = bench.gen_reads_device(codes, n_reads // 2, seed=args.seed, err=0.01, dev=dev)
del codes
ix = api.Index(prefix, 0, api.KG_SA_FULL if args.sa == "full" else api.KG_SA_SAMPLED)
enc_h, off_h = enc.cpu().numpy(), off.cpu().numpy()
ws = ix.workspace(n_reads, len(enc_h))
t = time.time(); so_g, s_g = ws.seed_batch(enc_h, off_h, 0); tg = time.time() - t
traffic = ws.traffic().as_dict()
orc = O.Oracle(prefix)
t = time.time(); so_o, s_o = orc.seed_batch(enc_h, off_h, 0, threads=bench.effective_cores()); to = time.time() - t
same = bool((so_g == so_o).all() and (s_g == s_o.astype(api.SEED_DT)).all())
# chaining on a prefix of the batch
k = min(args.cand_reads, n_reads)
cand_same = True
if k:
    wsk = api.Workspace(ix, k, k * bench.READ_LEN)
    so_k, _ = wsk.seed_batch(enc_h[: k * bench.READ_LEN], off_h[: k + 1], 0)
    got = wsk.candidates_batch(so_k, False)
    for r in range(k):
        want = orc.candidates(bench.READ_LEN, s_o[so_o[r]:so_o[r + 1]], False)
        if len(got[r]) != len(want) or any((gs, gp) != (ws_, wp) or len(gv) != len(wv) or not ((gv["gPos"] == wv["gPos"]).all() and (gv["rPos"] == wv["rPos"]).all() and (gv["len"] == wv["rLen"]).all())
                                           for (gs, gp, gv), (ws_, wp, wv) in zip(got[r], want)):
            cand_same = False
            break
    wsk.close()
print(json.dumps({"steps": {k_: traffic[k_] for k_ in ("rank_steps", "double_steps", "triple_steps")}, "genome_len": L, "reads": n_reads, "seeds": int(so_o[-1]), "seeds_identical": same, "candidate_reads": k, "candidates_identical": cand_same,
                  "sa": args.sa, "force_u64": bool(os.environ.get("KG_FORCE_U64")), "qmer": os.environ.get("KG_QMER"),
                  "gpu_host_form_s": round(tg, 2), "oracle_s": round(to, 1)}))
sys.exit(0 if same and cand_same else 1)

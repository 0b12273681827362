#!/usr/bin/env python3
"""Index (GPU-built, cached in --workdir) and paired-end FASTQ files on a large synthetic genome of bench.py's kind, for SAM parity
runs of kart-amd against the reference binary.  VALIDATION TOOL (GPU box); run by tests/test_large_gpu.py.
    python tools/large_sam_inputs.py --genome-len L --pairs N --workdir DIR [--err E]   -> JSON {"prefix", "f1", "f2"}"""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from kart_amd import index_build

ap = argparse.ArgumentParser()
ap.add_argument("--genome-len", type=int, default=50_000_000)
ap.add_argument("--pairs", type=int, default=100_000)
ap.add_argument("--workdir", required=True)
ap.add_argument("--err", type=float, default=0.01)
ap.add_argument("--seed", type=int, default=17)
args = ap.parse_args()
dev = torch.device("cuda", 0)
os.makedirs(args.workdir, exist_ok=True)
L = args.genome_len
prefix = os.path.join(args.workdir, "synth_v2_%d" % L)
codes = bench.make_large_codes(L, 3, dev)
if not all(os.path.exists(prefix + e) for e in (".bwt", ".sa", ".pac", ".ann", ".amb")):
    anns = [("decoy", "(null)", 0, bench.DECOY_LEN, 0)] + [(nm, "(null)", off, ln, 0) for nm, off, ln in bench.contig_table(L)]
    index_build.build_index_from_codes(codes.cpu().numpy(), anns, [], prefix, device=str(dev))
    torch.cuda.empty_cache()
f1, f2 = os.path.join(args.workdir, "sam_%d_1.fq" % args.pairs), os.path.join(args.workdir, "sam_%d_2.fq" % args.pairs)
bench.write_fastq_pairs(codes, args.pairs, args.seed, f1, f2, dev, err=args.err)
print(json.dumps({"prefix": prefix, "f1": f1, "f2": f2}))

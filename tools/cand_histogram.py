#!/usr/bin/env python3
"""How the candidates are spread over the read pairs of the bench workload (GPU box): seeds + chains 2 M reads of bench.py's generator on the
hg38-sized index and prints histograms of candidates per read, n1 * n2 per pair and seeds per read -- what the per-lane loops of aln_pair /
aln_final / chain_kernel run over.  DIAGNOSTIC.  usage: python tools/cand_histogram.py [pairs]"""
import argparse, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from kart_amd import api

pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = torch.device("cuda", 0)
args = argparse.Namespace(genome_len=bench.HG38_LEN, bucketed=None, repeat_frac=0.45)
work = bench.pick_workdir(12 << 30)
os.makedirs(work, exist_ok=True)
prefix, codes, _ = bench.prepare_index(args, dev, 0, work, lambda: None)
enc, off = bench.gen_reads_device(codes, pairs, seed=5, err=0.01, dev=dev)
bench.release_haplotypes()
enc_h, off_h = enc.cpu().numpy(), off.cpu().numpy()
del codes, enc, off
torch.cuda.empty_cache()
ix = api.Index(prefix, 0, api.KG_SA_AUTO)
n = 2 * pairs
ws = ix.workspace(n, len(enc_h))
so, seeds = ws.seed_batch(enc_h, off_h, api.KG_MODE_FAST)
spr = np.diff(so)
ncand = np.zeros(n + 1, dtype=np.int32)
import ctypes as C
pc, ps, nc, ns = C.c_void_p(), C.c_void_p(), C.c_int64(), C.c_int64()
api._check(ws.lib.kg_candidates_batch(ws.h, 0, 5, n, int(so[n]), api._ptr(ncand), C.byref(pc), C.byref(nc), C.byref(ps), C.byref(ns)), "kg_candidates_batch")
nc_r = ncand[:n].astype(np.int64)
prod = nc_r[0::2] * nc_r[1::2]


def hist(name, x, edges):
    tot = len(x)
    print(name, "mean %.2f max %d" % (x.mean(), x.max()))
    for lo, hi in zip(edges[:-1], edges[1:]):
        m = (x >= lo) & (x < hi)
        print("   [%5d, %5d)  %8.4f %% of items   %8.4f %% of the sum" % (lo, hi, 100.0 * m.sum() / tot, 100.0 * x[m].sum() / max(1, x.sum())))


edges = [0, 1, 2, 3, 5, 9, 17, 33, 65, 129, 257, 1025, 4097, 1 << 30]
hist("seeds per read", spr, edges)
hist("candidates per read", nc_r, edges)
hist("n1 * n2 per pair", prod, edges)
# a wave of 64 consecutive pairs costs its heaviest lane: the sum over waves of max(n1 * n2)
w = prod[:len(prod) // 64 * 64].reshape(-1, 64)
print("per wave of 64 consecutive pairs: mean of max(n1*n2) = %.1f, mean of mean = %.2f  (lock-step cost / useful work = %.1f x)" % (w.max(1).mean(), w.mean(), w.max(1).mean() / max(1e-9, w.mean())))
w = spr[:len(spr) // 64 * 64].reshape(-1, 64)
print("per wave of 64 consecutive reads: mean of max(seeds) = %.1f, mean of mean = %.2f (%.1f x)" % (w.max(1).mean(), w.mean(), w.max(1).mean() / max(1e-9, w.mean())))
ix.close()

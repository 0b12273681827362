#!/usr/bin/env python3
"""Kernel times of the long-read path (SensitiveMode seeding + PacBio chaining) on N x 7 kb reads (MEASUREMENT TOOL, GPU box)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from kart_amd import api, synth, index_build
import tempfile, os
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
d = tempfile.mkdtemp(prefix="kart_pbk")
genome = synth.make_genome([("decoy", 2000), ("chrE", 100_000_000)], seed=2, gc=0.508)
fa = os.path.join(d, "g.fa"); synth.write_fasta(fa, genome)
index_build.build_index(fa, os.path.join(d, "idx"))
names, reads = synth.simulate_long_reads(genome, n, seed=9, read_len=7000, err=0.15, indel_err_frac=0.1)
enc, off = api.concat_reads([synth.encode(r) for r in reads])
ix = api.Index(os.path.join(d, "idx"), 0, api.KG_SA_FULL)
ws = ix.workspace(len(off) - 1, len(enc))
ws.set_profiling(True)
for it in range(3):
    t = time.time(); so, seeds = ws.seed_batch(enc, off, 1); t1 = time.time() - t
    ms = ws.kernel_ms()
    import ctypes as C
    ncand = np.zeros(len(off), dtype=np.int32)
    pc, ps, nc, ns = C.c_void_p(), C.c_void_p(), C.c_int64(), C.c_int64()
    t = time.time(); rc = ws.lib.kg_candidates_batch(ws.h, 1, 5, len(off) - 1, int(so[-1]), ncand.ctypes.data, C.byref(pc), C.byref(nc), C.byref(ps), C.byref(ns)); t2 = time.time() - t
    print("seed_batch %.3f s (kernels ms: %s) | candidates_batch %.3f s rc=%d | seeds/read %.1f cands/read %.2f" % (t1, ms, t2, rc, so[-1] / (len(off) - 1), ncand.mean()))

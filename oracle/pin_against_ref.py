#!/usr/bin/env python3
"""Pin the CPU oracle against the UNMODIFIED reference and write the golden fixtures.

TEST INFRASTRUCTURE.  Runs only where /root/reference exists (it needs oracle/_ref, built by
`make -C oracle ref`).  For every function on the hot path it feeds the same seeded inputs to
  (a) oracle/liboracle.so       -- our restatement, and
  (b) oracle/_ref/libkartref.so -- the reference's own object code (through ref_shim.cpp),
asserts identical results, and stores inputs + the REFERENCE's outputs in tests/golden/ so the
pin can be re-checked (oracle vs golden, HIP vs golden) on machines without the reference.

    python oracle/pin_against_ref.py            # regenerate tests/golden/hotpath_small.npz + idx
"""
import os
import shutil
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kart_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def main():
    if not O.ref_available():
        O.build()
    assert O.ref_available(), "oracle/_ref is missing: run `make -C oracle ref` in a container with /root/reference"
    os.makedirs(os.path.join(GOLD, "idx"), exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="kartpin")
    # 103 kb genome: 3 kb leading decoy, three contigs, 10 % repeats, one N run
    genome = synth.make_genome([("decoy", 3000), ("chrA", 60000), ("chrB", 30000), ("chrC", 10000)], seed=1,
                               repeat_frac=0.10, n_runs=[("chrB", 12000, 40)])
    fa = os.path.join(GOLD, "small.fa")
    synth.write_fasta(fa, genome)
    prefix = os.path.join(GOLD, "idx", "small")
    O.ref_build_index(fa, prefix)
    for ext in (".bwt", ".sa", ".pac", ".ann", ".amb"):
        assert os.path.exists(prefix + ext)

    orc = O.Oracle(prefix)
    ref = O.RefShim(prefix)
    assert orc.genome_size == ref.genome_size and orc.primary == ref.primary and orc.min_seed_len == ref.min_seed_len
    assert (orc.ref_sequence() == ref.ref_sequence()).all()
    out = {}
    rng = np.random.default_rng(7)
    N = orc.seq_len

    # ---- rank / SA known-answer tests --------------------------------------------------------------
    ks = np.concatenate([rng.integers(0, N + 1, size=600), [0, N, orc.primary, orc.primary - 1, orc.primary + 1, 127, 128, 129]]).astype(np.uint64)
    occ = np.zeros((len(ks), 4), dtype=np.uint64)
    sa = np.zeros(len(ks), dtype=np.uint64)
    for i, k in enumerate(ks):
        k = int(k)
        r4 = ref.occ4(k)
        assert (orc.occ4(k) == r4).all()
        for c in range(4):
            assert orc.occ(k, c) == ref.occ(k, c) == int(r4[c]), (k, c)
        occ[i] = r4
        sa[i] = ref.sa(k)
        assert orc.sa(k) == int(sa[i])
    out.update(kat_k=ks, kat_occ4=occ, kat_sa=sa)

    # ---- seeding, FastMode: 300 pairs of 150 bp (1 % errors, a few N) + ragged/edge reads -----------
    _, r1, r2 = synth.simulate_pairs(genome, 300, seed=3, n_frac=0.002)
    reads = [synth.encode(r) for r in r1] + [synth.encode(synth.revcomp(r)) for r in r2]
    chrA = genome["chrA"]
    edge = [chrA[100:100 + L] for L in (1, 12, 13, 14, 15, 27, 28, 40, 151, 400)]
    edge += [np.full(50, ord("N"), np.uint8), np.frombuffer(b"ACGT" * 20, dtype=np.uint8)]
    edge += [genome["decoy"][0:150], synth.revcomp(genome["chrC"][-150:]), genome["chrB"][11950:12100]]
    reads += [synth.encode(e) for e in edge]
    fast_enc, fast_off = _concat(reads)
    fast_seeds, fast_so = [], [0]
    for e in reads:
        a, b = orc.seed_read(e, 0), ref.seed_read(e, 0)
        assert len(a) == len(b) and (a == b).all()
        fast_seeds.append(b)
        fast_so.append(fast_so[-1] + len(b))
    out.update(fast_enc=fast_enc, fast_off=fast_off, fast_seed_off=np.array(fast_so, dtype=np.int64),
               fast_seeds=np.concatenate(fast_seeds))

    # ---- seeding, SensitiveMode: 24 long reads (7 % subs) + the short ones again ---------------------
    _, longs = synth.simulate_long_reads(genome, 24, seed=5, read_len=3000, err=0.07)
    sreads = [synth.encode(r) for r in longs] + reads[:60] + reads[-15:]
    sreads[3] = sreads[3].copy(); sreads[3][500:520] = 4   # an N run inside a long read (App. B-10 path)
    sreads[4] = sreads[4].copy(); sreads[4][-45:-20] = 4   # ... and one near the read end
    sens_enc, sens_off = _concat(sreads)
    sens_seeds, sens_so = [], [0]
    for e in sreads:
        a, b = orc.seed_read(e, 1), ref.seed_read(e, 1)
        assert len(a) == len(b) and (a == b).all()
        sens_seeds.append(b)
        sens_so.append(sens_so[-1] + len(b))
    out.update(sens_enc=sens_enc, sens_off=sens_off, sens_seed_off=np.array(sens_so, dtype=np.int64),
               sens_seeds=np.concatenate(sens_seeds))
    out["counters_oracle"] = np.array(list(orc.counters().values()), dtype=np.uint64)

    # ---- NW: random and near-identical pairs, lengths 1..200, N and lower case included --------------
    pairs = []
    alpha = np.frombuffer(b"ACGTacgtN", dtype=np.uint8)
    for it in range(1500):
        cls = it % 5
        hi = (9, 9, 33, 33, 200)[cls]
        m = int(rng.integers(1, hi))
        a = alpha[rng.integers(0, 4 if it % 7 else 9, size=m)].tobytes()
        if rng.random() < 0.7:
            bb = bytearray(a)
            for _ in range(int(rng.integers(0, 4))):
                p = int(rng.integers(0, len(bb) + 1))
                if rng.random() < 0.5 and len(bb) > 1:
                    del bb[min(p, len(bb) - 1)]
                else:
                    bb.insert(p, int(alpha[rng.integers(0, 4)]))
            if rng.random() < 0.5 and len(bb) > 0:
                p = int(rng.integers(0, len(bb)))
                bb[p] = int(alpha[rng.integers(0, 4)])
            b = bytes(bb)
        else:
            b = alpha[rng.integers(0, 5, size=int(rng.integers(1, hi)))].tobytes()
        pairs.append((a, b))
    pairs += [(b"A", b"A"), (b"A", b"C"), (b"A", b"ACGTACGT"), (b"ACGTACGT", b"T"), (b"N", b"n"), (b"acgt", b"ACGT")]
    nw1, nw2 = [], []
    for a, b in pairs:
        x, y = orc.nw(a, b), ref.nw(a, b)
        assert x == y, (a, b, x, y)
        nw1.append(y[0]); nw2.append(y[1])
    out.update(nw_s1=np.array([p[0] for p in pairs], dtype=object), nw_s2=np.array([p[1] for p in pairs], dtype=object),
               nw_a1=np.array(nw1, dtype=object), nw_a2=np.array(nw2, dtype=object))

    # ---- chaining + normal pairs ---------------------------------------------------------------------
    cand_rows = []   # (mode, read idx, score, posdiff, n_pairs)
    cand_pairs, np_counts, np_pairs = [], [], []
    for pacbio, rlist, seedlist in ((False, reads, fast_seeds), (True, sreads, sens_seeds)):
        ref.set_mode(pacbio)
        for ri, (e, s) in enumerate(zip(rlist, seedlist)):
            ca, cb = orc.candidates(len(e), s, pacbio), ref.candidates(len(e), s, pacbio)
            assert len(ca) == len(cb)
            for (s1, p1, v1), (s2, p2, v2) in zip(ca, cb):
                assert s1 == s2 and p1 == p2 and (v1 == v2).all()
                n1, n2 = orc.identify_normal_pairs(len(e), -1, v1), ref.identify_normal_pairs(len(e), -1, v2)
                assert len(n1) == len(n2)
                for f in ("gPos", "rPos", "rLen", "gLen", "bSimple"):   # PosDiff of added pairs is stale in the reference
                    assert (n1[f] == n2[f]).all(), (f, n1, n2)
                cand_rows.append((int(pacbio), ri, s2, p2, len(v2)))
                cand_pairs.append(v2)
                np_counts.append(len(n2))
                np_pairs.append(n2)
    out.update(cand_rows=np.array(cand_rows, dtype=np.int64), cand_pairs=np.concatenate(cand_pairs),
               np_counts=np.array(np_counts, dtype=np.int64), np_pairs=np.concatenate(np_pairs))

    np.savez_compressed(os.path.join(GOLD, "hotpath_small.npz"), **out)
    shutil.rmtree(tmp, ignore_errors=True)
    print("pinned: oracle == reference on", len(ks), "rank/SA KATs,", len(reads), "+", len(sreads), "reads,",
          len(pairs), "NW pairs,", len(cand_rows), "candidates; fixtures written to", GOLD)


def _concat(reads):
    off = np.zeros(len(reads) + 1, dtype=np.int64)
    np.cumsum([len(r) for r in reads], out=off[1:])
    return np.concatenate(reads).astype(np.uint8), off


if __name__ == "__main__":
    main()

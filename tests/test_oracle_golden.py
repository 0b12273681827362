"""CPU: the oracle restatement reproduces the reference's outputs stored in tests/golden/
(written by oracle/pin_against_ref.py from the unmodified reference's object code)."""
import numpy as np

from conftest import split
from oracle import oracle as O


def test_rank_and_sa_kats(golden, oracle_small):
    for k, occ4, sa in zip(golden["kat_k"], golden["kat_occ4"], golden["kat_sa"]):
        k = int(k)
        assert (oracle_small.occ4(k) == occ4).all()
        for c in range(4):
            assert oracle_small.occ(k, c) == int(occ4[c])
        assert oracle_small.sa(k) == int(sa)


def test_index_constants(oracle_small):
    assert oracle_small.genome_size == 103000
    assert oracle_small.seq_len == 206000
    assert oracle_small.min_seed_len == 13   # 2L < 4^13 (reference src/Mapping.cpp:645)
    assert oracle_small.n_contigs == 4


def test_seeds_fast_mode(golden, oracle_small):
    reads = split(golden["fast_enc"], golden["fast_off"])
    want = split(golden["fast_seeds"], golden["fast_seed_off"])
    for e, w in zip(reads, want):
        got = oracle_small.seed_read(e, 0)
        assert len(got) == len(w) and (got == w.astype(O.SEED_DT)).all()


def test_seeds_sensitive_mode(golden, oracle_small):
    reads = split(golden["sens_enc"], golden["sens_off"])
    want = split(golden["sens_seeds"], golden["sens_seed_off"])
    for e, w in zip(reads, want):
        got = oracle_small.seed_read(e, 1)
        assert len(got) == len(w) and (got == w.astype(O.SEED_DT)).all()


def test_seed_batch_equals_per_read_and_threads(golden, oracle_small):
    so1, s1 = oracle_small.seed_batch(golden["fast_enc"], golden["fast_off"], 0, threads=1)
    so3, s3 = oracle_small.seed_batch(golden["fast_enc"], golden["fast_off"], 0, threads=3)
    assert (so1 == golden["fast_seed_off"]).all() and (so3 == so1).all()
    assert (s1 == golden["fast_seeds"].astype(O.SEED_DT)).all() and (s3 == s1).all()


def test_seeds_are_exact_matches(golden, oracle_small):
    """size-independent property: every seed is an exact match of the read in the 2L text"""
    ref = oracle_small.ref_sequence()
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    reads = split(golden["fast_enc"], golden["fast_off"])
    seeds = split(golden["fast_seeds"], golden["fast_seed_off"])
    n = 0
    for e, ss in zip(reads, seeds):
        for s in ss:
            frag = e[s["rPos"]:s["rPos"] + s["len"]]
            assert (frag <= 3).all()
            assert (acgt[frag] == ref[s["gPos"]:s["gPos"] + s["len"]]).all()
            n += 1
    assert n > 500


def test_nw(golden, oracle_small):
    for a, b, x, y in zip(golden["nw_s1"], golden["nw_s2"], golden["nw_a1"], golden["nw_a2"]):
        assert oracle_small.nw(a, b) == (x, y)


def test_chaining_and_normal_pairs(golden, oracle_small):
    fast_reads = split(golden["fast_enc"], golden["fast_off"])
    fast_seeds = split(golden["fast_seeds"], golden["fast_seed_off"])
    sens_reads = split(golden["sens_enc"], golden["sens_off"])
    sens_seeds = split(golden["sens_seeds"], golden["sens_seed_off"])
    rows = golden["cand_rows"]
    cp_off = np.concatenate([[0], np.cumsum(rows[:, 4])])
    np_off = np.concatenate([[0], np.cumsum(golden["np_counts"])])
    gi = 0
    for pacbio, rlist, slist in ((0, fast_reads, fast_seeds), (1, sens_reads, sens_seeds)):
        for ri, (e, s) in enumerate(zip(rlist, slist)):
            for score, pd, v in oracle_small.candidates(len(e), s.astype(O.SEED_DT), bool(pacbio)):
                assert tuple(rows[gi][:4]) == (pacbio, ri, score, pd)
                want = golden["cand_pairs"][cp_off[gi]:cp_off[gi + 1]]
                assert (v == want.astype(O.PAIR_DT)).all()
                got_np = oracle_small.identify_normal_pairs(len(e), -1, v)
                want_np = golden["np_pairs"][np_off[gi]:np_off[gi + 1]]
                assert len(got_np) == len(want_np)
                for f in ("gPos", "rPos", "rLen", "gLen", "bSimple"):
                    assert (got_np[f] == want_np[f]).all()
                gi += 1
    assert gi == len(rows)


def test_normal_pair_alignment_golden():
    """GenerateNormalPairAlignment (8-mer partition, IdentifyNormalPairs on the fragment, NW, the -pacbio recursion): the oracle against
    the reference's aligned strings for 267 fragment pairs in both modes (oracle/pin_fragments_against_ref.py)"""
    import os
    import numpy as np
    from conftest import ROOT, SMALL_PREFIX
    from oracle import oracle as O
    g = np.load(os.path.join(ROOT, "tests", "golden", "fragments_small.npz"), allow_pickle=True)
    orc = O.Oracle(SMALL_PREFIX)
    text = orc.ref_sequence()
    for mode, pacbio in (("pacbio", True), ("illumina", False)):
        for i in range(len(g["frag1"])):
            gp, gl = int(g["gpos"][i]), int(g["glen"][i])
            got = orc.normal_pair_alignment(bytes(g["frag1"][i]), text[gp:gp + gl].tobytes(), pacbio, 5)
            assert got == (bytes(g["aln1_" + mode][i]), bytes(g["aln2_" + mode][i])), (mode, i)
    orc.close()

"""kg_fragments_batch (GenerateNormalPairAlignment on the device: frag_kernels.hip, abi_frag.hip) through the C ABI against the golden
fixture tests/golden/fragments_small.npz -- the UNMODIFIED reference's aligned strings for 267 fragment pairs in both modes, written
by oracle/pin_fragments_against_ref.py (reference src/tools.cpp:142-223, src/KmerAnalysis.cpp:56-179) -- and against the CPU oracle on
fresh random pairs.  Bit-exact: the op string must reproduce both aligned strings."""
import os

import numpy as np
import pytest

from conftest import ROOT
from kart_amd import api

pytestmark = pytest.mark.gpu
GOLD = os.path.join(ROOT, "tests", "golden", "fragments_small.npz")


def _expected_status(r, glen):
    # outside the kernels' envelope: both sides above 30 (src/tools.cpp:146) AND a read character other than A/C/G/T or a side above 8192 (kFragMaxLen)
    plain = all(c in b"ACGTacgt" for c in r)
    return 1 if (len(r) > 30 and glen > 30 and (not plain or len(r) > 8192 or glen > 8192)) else 0


@pytest.mark.parametrize("mode", ["pacbio", "illumina"])
def test_fragments_golden(gpu_index_full, oracle_small, mode):
    g = np.load(GOLD, allow_pickle=True)
    text = oracle_small.ref_sequence()
    frags = [bytes(x) for x in g["frag1"]]
    got, status = gpu_index_full.GenerateNormalPairAlignment(frags, g["gpos"], g["glen"], text, pacbio=(mode == "pacbio"), max_gaps=5)
    served = 0
    for i, r in enumerate(frags):
        assert int(status[i]) == _expected_status(r, int(g["glen"][i])), (i, len(r), int(g["glen"][i]))
        if status[i]:
            continue
        served += 1
        assert got[i] == (bytes(g["aln1_" + mode][i]), bytes(g["aln2_" + mode][i])), (mode, i, len(r), int(g["glen"][i]))
    assert served > 200


def test_fragments_random_vs_oracle(gpu_index_full, oracle_small):
    """fresh pairs at PacBio error rates, lengths up to the kernels' limit and beyond it, whole batch in one call"""
    rng = np.random.default_rng(99)
    text = oracle_small.ref_sequence()
    L = len(text) // 2
    frags, gpos, glen = [], [], []
    acgt = np.frombuffer(b"ACGT", np.uint8)
    for it in range(400):
        gl = int(rng.integers(1, (60, 400, 1500, 4300, 9000)[it % 5]))
        gp = int(rng.integers(3000, L - gl - 1))
        gseq = text[gp:gp + gl]
        if (gseq == ord("N")).any():
            continue
        keep = rng.random(gl) > 0.04
        r = gseq[keep].copy()
        e = rng.random(len(r)) < 0.12
        r[e] = acgt[rng.integers(0, 4, int(e.sum()))]
        ins = np.sort(rng.integers(0, len(r) + 1, size=max(0, len(r) // 40)))
        r = np.insert(r, ins, acgt[rng.integers(0, 4, len(ins))])
        if len(r) == 0:
            continue
        frags.append(r.tobytes()); gpos.append(gp); glen.append(gl)
    got, status = gpu_index_full.GenerateNormalPairAlignment(frags, gpos, glen, text, pacbio=True)
    for i, r in enumerate(frags):
        # (a fragment of several thousand noisy bases may also exceed the kernel's 383 exact matches: handed back as well; whatever IS served must be right)
        if max(len(r), glen[i]) <= 4096 or _expected_status(r, glen[i]):
            assert int(status[i]) == _expected_status(r, glen[i]), (i, len(r), glen[i])
        if not status[i]:
            assert got[i] == oracle_small.normal_pair_alignment(r, text[gpos[i]:gpos[i] + glen[i]].tobytes(), True, 5), (i, len(r), glen[i])
    assert (np.asarray(status) == 1).any() and (np.asarray(status) == 0).sum() > 300


def test_fragments_argument_errors(gpu_index_full):
    ix = gpu_index_full
    lib = ix.lib
    f1 = np.frombuffer(b"ACGTACGTAC" + b"\0" * 64, np.uint8).copy()
    off = np.array([0, 10], np.int64); g = np.array([5000], np.int64); gl = np.array([10], np.int32); oo = np.array([0], np.int64)
    ops = np.zeros(128, np.uint8); alen = np.zeros(1, np.int32); st = np.zeros(1, np.uint8)
    P = api._ptr
    assert lib.kg_fragments_batch(ix.h, P(f1), P(off), P(g), P(gl), 1, 1, 5, P(ops), P(oo), P(alen), P(st)) == api.KG_OK and 10 <= alen[0] <= 20
    assert lib.kg_fragments_batch(ix.h, P(f1), P(off), P(g), P(gl), 0, 1, 5, P(ops), P(oo), P(alen), P(st)) == api.KG_OK           # empty batch
    bad_g = np.array([1 << 40], np.int64)
    assert lib.kg_fragments_batch(ix.h, P(f1), P(off), P(bad_g), P(gl), 1, 1, 5, P(ops), P(oo), P(alen), P(st)) != api.KG_OK       # outside the text
    bad_oo = np.array([7], np.int64)
    assert lib.kg_fragments_batch(ix.h, P(f1), P(off), P(g), P(gl), 1, 1, 5, P(ops), P(bad_oo), P(alen), P(st)) != api.KG_OK       # ops_off is not the running column count
    assert lib.kg_fragments_batch(None, P(f1), P(off), P(g), P(gl), 1, 1, 5, P(ops), P(oo), P(alen), P(st)) != api.KG_OK
    assert lib.kg_fragments_batch(ix.h, None, P(off), P(g), P(gl), 1, 1, 5, P(ops), P(oo), P(alen), P(st)) != api.KG_OK

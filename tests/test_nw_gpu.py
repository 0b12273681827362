"""GPU parity for the gap-closing path: kg_nw_batch vs the reference's golden alignments and vs
the CPU oracle; integer DP must reproduce the float reference byte for byte."""
import numpy as np
import pytest

from kart_amd import api

pytestmark = pytest.mark.gpu


def test_nw_golden(golden, gpu_index):
    pairs = list(zip(golden["nw_s1"], golden["nw_s2"]))
    got = gpu_index.nw_alignment(pairs)
    for (a, b), (x, y), gx, gy in zip(pairs, got, golden["nw_a1"], golden["nw_a2"]):
        assert (x, y) == (gx, gy), (a, b)


def test_nw_size_classes_vs_oracle(gpu_index, oracle_small):
    """all three kernels: <=8 (registers), <=32 (LDS), >32 (wave sweep incl. multi-stripe)"""
    rng = np.random.default_rng(21)
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    pairs = []
    for m, n in [(1, 1), (8, 8), (9, 3), (32, 32), (33, 1), (1, 33), (64, 64), (65, 64), (64, 65), (130, 127), (200, 310), (700, 650)]:
        for rep in range(6):
            a = alpha[rng.integers(0, 4, size=m)]
            if rep % 2 == 0 and m > 4:
                b = list(a)
                for _ in range(max(1, m // 15)):
                    p = int(rng.integers(0, len(b)))
                    r = rng.random()
                    if r < 0.3 and len(b) > 1:
                        del b[p]
                    elif r < 0.6:
                        b.insert(p, int(alpha[rng.integers(0, 4)]))
                    else:
                        b[p] = int(alpha[rng.integers(0, 4)])
                b = np.array(b, dtype=np.uint8)
                b = np.resize(b, n) if len(b) != n and rep % 4 == 0 else b
            else:
                b = alpha[rng.integers(0, 4, size=n)]
            pairs.append((a.tobytes(), b.tobytes()))
    got = gpu_index.nw_alignment(pairs)
    for (a, b), g in zip(pairs, got):
        assert g == oracle_small.nw(a, b), (len(a), len(b))


def test_nw_properties_large_batch(gpu_index):
    """size-independent checks on 200k tiny pairs: ops consume both fragments exactly, and
    identical fragments align without gaps"""
    rng = np.random.default_rng(3)
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    pairs = []
    for i in range(200000):
        m = int(rng.integers(1, 9))
        a = alpha[rng.integers(0, 4, size=m)].tobytes()
        pairs.append((a, a) if i % 3 == 0 else (a, alpha[rng.integers(0, 4, size=int(rng.integers(1, 9)))].tobytes()))
    ops = gpu_index.nw_ops(pairs)
    for i, ((a, b), o) in enumerate(zip(pairs, ops)):
        assert int((o != api.KG_OP_GAP1).sum()) == len(a) and int((o != api.KG_OP_GAP2).sum()) == len(b)
        if i % 3 == 0:
            assert (o == api.KG_OP_DIAG).all()


def test_nw_empty_batch(gpu_index):
    assert gpu_index.nw_ops([]) == []


def test_nw_long_fragments_up_to_the_limit(gpu_index, oracle_small):
    """wave-per-pair kernel over many 64-column stripes, direction words in the HBM slab: 2.5 k, 5 k and the 7000-base limit,
    related sequences with indels (the long-read case) and one unrelated pair"""
    rng = np.random.default_rng(77)
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    pairs = []
    for m in (2500, 5000, 7000):
        a = alpha[rng.integers(0, 4, size=m)]
        b = list(a)
        for _ in range(m // 12):
            p = int(rng.integers(0, len(b)))
            r = rng.random()
            if r < 0.35 and len(b) > 1:
                del b[p]
            elif r < 0.7 and len(b) < 6999:
                b.insert(p, int(alpha[rng.integers(0, 4)]))
            else:
                b[p] = int(alpha[rng.integers(0, 4)])
        pairs.append((a.tobytes(), np.array(b[:7000], dtype=np.uint8).tobytes()))
    pairs.append((alpha[rng.integers(0, 4, size=3000)].tobytes(), alpha[rng.integers(0, 4, size=2700)].tobytes()))
    got = gpu_index.nw_alignment(pairs)
    for (a, b), g in zip(pairs, got):
        assert g == oracle_small.nw(a, b), (len(a), len(b))


def test_nw_fragments_longer_than_the_lds_holds(gpu_index, oracle_small):
    """> 7000 bases: the wave-per-pair sweep with its boundary column and sequence codes in the wave's HBM slab instead of the
    LDS (the reference allocates full matrices of any size, src/nw_alignment.cpp:24-33): 7001 x 7001 related, 9000 x 8200
    with indels, a long fragment against a short one and the reverse, mixed with small pairs of the other size classes"""
    rng = np.random.default_rng(79)
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)

    def mutate(a, k):
        b = list(a)
        for _ in range(k):
            p = int(rng.integers(0, len(b)))
            r = rng.random()
            if r < 0.35 and len(b) > 1:
                del b[p]
            elif r < 0.7:
                b.insert(p, int(alpha[rng.integers(0, 4)]))
            else:
                b[p] = int(alpha[rng.integers(0, 4)])
        return np.array(b, dtype=np.uint8).tobytes()

    a1 = alpha[rng.integers(0, 4, size=7001)]
    a2 = alpha[rng.integers(0, 4, size=9000)]
    pairs = [(a1.tobytes(), mutate(a1, 300)[:7001]), (a2.tobytes(), mutate(a2[:8200], 500)), (a2.tobytes()[:7500], b"ACGTTGCA" * 5),
             (b"ACGTTGCAAC" * 4, a2.tobytes()[:7300]), (b"ACGT", b"ACT"), (b"ACGTACGTACGTACGTAAAC", b"ACGTACGTCGTACGTAAAC"), (a1.tobytes()[:100], a1.tobytes()[3:90])]
    got = gpu_index.nw_alignment(pairs)
    for (a, b), g in zip(pairs, got):
        assert g == oracle_small.nw(a, b), (len(a), len(b))

#!/usr/bin/env python3
"""Pin the oracle's GenerateNormalPairAlignment (8-mer partition, IdentifyNormalPairs on the fragment, NW of the pieces, the -pacbio
recursion; reference src/tools.cpp:142-223, src/KmerAnalysis.cpp:56-179) against the UNMODIFIED reference and write the golden fixture.

TEST INFRASTRUCTURE.  Runs only where /root/reference exists (oracle/_ref, `make -C oracle ref`).  Seeded fragment pairs -- a window
of the small golden genome against a mutated copy of it (substitutions, insertions, deletions), unrelated pairs, lengths around the
30 / 300 thresholds, 'N', lower case and other characters in the read fragment -- go through liboracle.so and through the reference's
own object code in both modes (-pacbio and not); the inputs and the REFERENCE's aligned strings are stored in
tests/golden/fragments_small.npz.

    python oracle/pin_fragments_against_ref.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def mutate(rng, s, sub, ins, dele):
    alpha = b"ACGT"
    out = bytearray()
    for ch in s:
        r = rng.random()
        if r < dele:
            continue
        if r < dele + sub:
            out.append(alpha[(alpha.index(ch) + int(rng.integers(1, 4))) % 4] if ch in alpha else alpha[int(rng.integers(0, 4))])
        else:
            out.append(ch)
        if rng.random() < ins:
            out.append(alpha[int(rng.integers(0, 4))])
    return bytes(out) or b"A"


def make_cases(ref_seq, rng):
    """(read fragment, gpos, glen): the genome fragment is the reference text itself, forward strand, as in the callers (src/tools.cpp:259,292,336)"""
    L = len(ref_seq) // 2
    cases = []
    lens = [31, 32, 40, 64, 100, 150, 250, 299, 300, 301, 302, 350, 500, 800, 1200, 2000, 3000, 4000]
    for it in range(260):
        glen = int(lens[it % len(lens)] + rng.integers(0, 7))
        gpos = int(rng.integers(3000, L - glen - 1))
        g = ref_seq[gpos:gpos + glen].tobytes()
        if b"N" in g:
            continue
        kind = it % 13
        if kind == 12:
            r = bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=glen))          # unrelated: no common 8-mer, or a few by chance
        else:
            r = mutate(rng, g, sub=(0.12, 0.05, 0.2)[it % 3], ins=0.02, dele=0.02)
        if kind == 5:
            r = r[: len(r) // 2] + mutate(rng, g[len(g) // 3:], 0.1, 0.02, 0.02)         # a repeated stretch: tandem / translocated seeds
        if kind == 7 and len(r) > 60:
            p = int(rng.integers(10, len(r) - 10)); r = r[:p] + b"N" + r[p + 1:]
        if kind == 9:
            r = r.lower()
        if kind == 10 and len(r) > 60:
            p = int(rng.integers(10, len(r) - 10)); r = r[:p] + b"R" + r[p + 1:]         # an IUPAC code: nst_nt4_table gives 4, not skipped like 'N'
        cases.append((r, gpos, glen))
    # the thresholds of :146 (both sides above 30) and short sides
    for rl, gl in ((30, 200), (200, 30), (31, 31), (1, 50), (50, 1), (29, 29), (8, 8)):
        gpos = int(rng.integers(3000, L - gl - 1))
        g = ref_seq[gpos:gpos + gl].tobytes()
        r = mutate(rng, ref_seq[gpos:gpos + max(rl, 1)].tobytes(), 0.1, 0.0, 0.0)[:rl].ljust(rl, b"A")
        cases.append((r, gpos, gl))
    return cases


def main():
    assert O.ref_available(), "oracle/_ref is missing: run `make -C oracle ref` in a container with /root/reference"
    prefix = os.path.join(GOLD, "idx", "small")
    orc, ref = O.Oracle(prefix), O.RefShim(prefix)
    ref_seq = orc.ref_sequence()
    rng = np.random.default_rng(2024)
    cases = make_cases(ref_seq, rng)
    out = {"frag1": np.array([c[0] for c in cases], dtype=object), "gpos": np.array([c[1] for c in cases], dtype=np.int64), "glen": np.array([c[2] for c in cases], dtype=np.int32)}
    for pacbio, max_gaps, tag in ((True, 5, "pacbio"), (False, 5, "illumina")):
        ref.set_mode(pacbio, max_gaps)
        a1, a2 = [], []
        for r, gpos, glen in cases:
            g = ref_seq[gpos:gpos + glen].tobytes()
            x, y = orc.normal_pair_alignment(r, g, pacbio, max_gaps), ref.normal_pair_alignment(r, g)
            assert x == y, (tag, len(r), glen, r[:60], x[0][:80], y[0][:80])
            a1.append(y[0]); a2.append(y[1])
        out["aln1_" + tag] = np.array(a1, dtype=object)
        out["aln2_" + tag] = np.array(a2, dtype=object)
    np.savez_compressed(os.path.join(GOLD, "fragments_small.npz"), **out)
    print("pinned: oracle == reference on", len(cases), "fragment pairs x 2 modes; fixture written to", os.path.join(GOLD, "fragments_small.npz"))


if __name__ == "__main__":
    main()

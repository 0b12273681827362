"""Seeded synthetic genomes and wgsim-style reads (numpy, host side).

The GPU box has no E. coli / hg38 FASTA and the reference's bundled read simulator
is time-seeded (reference wgsim/wgsim.c:447-449), so every test and benchmark
input is generated here from an explicit seed.  The read
model follows the bundled wgsim's: outer distance ~ N(500, 50), substitution
sequencing errors with its recurrent replacement rule (wgsim.c:368), haplotype
mutations at rate 0.001 of which 15 % are single-base indels (wgsim.c:98-102),
mate 2 reported as the reverse complement of the far end of the fragment.
"""
from __future__ import annotations

import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
_COMP[:] = ord("N")
for _a, _b in zip(b"ACGTacgt", b"TGCATGCA"):
    _COMP[_a] = _b


def revcomp(a: np.ndarray) -> np.ndarray:
    """Reverse complement along the last axis of an ASCII uint8 array."""
    return _COMP[a[..., ::-1]]


def make_genome(contigs, seed: int, gc: float = 0.5, repeat_frac: float = 0.0,
                repeat_len: int = 300, repeat_families: int = 4, repeat_div: float = 0.02,
                n_runs=()):
    """Return an ordered dict name -> ASCII uint8 array.

    contigs: iterable of (name, length).  repeat_frac of every contig (except one
    called 'decoy') is overwritten with mutated copies of a few repeat families so
    that multi-hit seeds and freq>50 drops occur.  n_runs: iterable of
    (contig_name, start, length) stretches replaced by 'N'.
    """
    rng = np.random.default_rng(seed)
    p = np.array([(1 - gc) / 2, gc / 2, gc / 2, (1 - gc) / 2])
    fams = [_ACGT[rng.choice(4, size=repeat_len, p=p)] for _ in range(repeat_families)]
    out = {}
    for name, length in contigs:
        seq = _ACGT[rng.choice(4, size=int(length), p=p)]
        if repeat_frac > 0 and name != "decoy" and length > 4 * repeat_len:
            n_copies = int(length * repeat_frac / repeat_len)
            starts = rng.integers(0, length - repeat_len, size=n_copies)
            for s in starts:
                fam = fams[int(rng.integers(repeat_families))].copy()
                mut = rng.random(repeat_len) < repeat_div
                fam[mut] = _ACGT[rng.integers(0, 4, size=int(mut.sum()))]
                seq[s:s + repeat_len] = fam
        out[name] = seq
    for name, start, length in n_runs:
        out[name][start:start + length] = ord("N")
    return out


def write_fasta(path, genome, width: int = 60):
    with open(path, "wb") as fh:
        for name, seq in genome.items():
            fh.write(b">" + name.encode() + b"\n")
            n = len(seq)
            body = seq.tobytes()
            for i in range(0, n, width):
                fh.write(body[i:i + width] + b"\n")


def _mutate_haplotype(seq: np.ndarray, rng, rate: float, indel_frac: float) -> np.ndarray:
    """Apply substitutions and 1-bp indels at `rate`; returns the mutated copy."""
    if rate <= 0:
        return seq
    n = len(seq)
    hit = np.nonzero(rng.random(n) < rate)[0]
    if len(hit) == 0:
        return seq
    kind = rng.random(len(hit))
    out = seq.copy()
    subs = hit[kind >= indel_frac]
    out[subs] = _ACGT[rng.integers(0, 4, size=len(subs))]
    dels = hit[kind < indel_frac / 2]
    ins = hit[(kind >= indel_frac / 2) & (kind < indel_frac)]
    keep = np.ones(n, dtype=bool)
    keep[dels] = False
    rep = np.ones(n, dtype=np.int64)
    rep[ins] = 2
    rep[~keep] = 0
    out = np.repeat(out, rep)
    return out


def simulate_pairs(genome, n_pairs: int, seed: int, read_len: int = 150, err: float = 0.01,
                   mut: float = 0.001, indel_frac: float = 0.15, ins_mean: float = 500.0,
                   ins_sd: float = 50.0, skip=("decoy",), n_frac: float = 0.0):
    """Return (names, r1, r2): r1/r2 are (n_pairs, read_len) ASCII arrays as they
    would appear in the two FASTQ files (mate 2 is the reverse complement of the
    fragment's far end, as a sequencer reports it)."""
    rng = np.random.default_rng(seed)
    names_c = [c for c in genome if c not in skip]
    haps = {c: _mutate_haplotype(genome[c], rng, mut, indel_frac) for c in names_c}
    lens = np.array([len(haps[c]) for c in names_c], dtype=np.int64)
    usable = lens > int(ins_mean + 6 * ins_sd) + read_len
    w = np.where(usable, lens, 0).astype(np.float64)
    w /= w.sum()
    which = rng.choice(len(names_c), size=n_pairs, p=w)
    frag = np.clip(rng.normal(ins_mean, ins_sd, size=n_pairs).round().astype(np.int64), read_len, None)
    r1 = np.empty((n_pairs, read_len), dtype=np.uint8)
    r2 = np.empty((n_pairs, read_len), dtype=np.uint8)
    pos = np.empty(n_pairs, dtype=np.int64)
    ar = np.arange(read_len)
    for ci, cname in enumerate(names_c):
        sel = np.nonzero(which == ci)[0]
        if len(sel) == 0:
            continue
        h = haps[cname]
        p0 = (rng.random(len(sel)) * (len(h) - frag[sel])).astype(np.int64)
        pos[sel] = p0
        left = h[p0[:, None] + ar]
        right = revcomp(h[(p0 + frag[sel] - read_len)[:, None] + ar])
        flip = rng.random(len(sel)) < 0.5
        r1[sel] = np.where(flip[:, None], right, left)
        r2[sel] = np.where(flip[:, None], left, right)
    for arr in (r1, r2):
        e = rng.random(arr.shape) < err
        code = (np.searchsorted(_ACGT, arr[e]) + 1 + rng.integers(0, 3, size=int(e.sum()))) & 3
        # a stored 'N' (from an N run) stays as it is
        repl = _ACGT[code]
        keep_n = arr[e] == ord("N")
        repl[keep_n] = ord("N")
        arr[e] = repl
        if n_frac > 0:
            arr[rng.random(arr.shape) < n_frac] = ord("N")
    names = [f"r{i}:Pos={int(pos[i])}:{names_c[int(which[i])]}" for i in range(n_pairs)]
    return names, r1, r2


def simulate_long_reads(genome, n_reads: int, seed: int, read_len: int = 7000, err: float = 0.15,
                        indel_err_frac: float = 0.0, skip=("decoy",)):
    """Single-end long reads with substitution errors (wgsim -e) and, optionally, a share of
    the errors turned into 1-bp insertions/deletions.  Returns (names, list of ASCII arrays)."""
    rng = np.random.default_rng(seed)
    names_c = [c for c in genome if c not in skip and len(genome[c]) > read_len + 10]
    lens = np.array([len(genome[c]) for c in names_c], dtype=np.float64)
    which = rng.choice(len(names_c), size=n_reads, p=lens / lens.sum())
    names, reads = [], []
    for i in range(n_reads):
        g = genome[names_c[int(which[i])]]
        p0 = int(rng.integers(0, len(g) - read_len))
        r = g[p0:p0 + read_len].copy()
        if rng.random() < 0.5:
            r = revcomp(r)
        e = rng.random(read_len) < err
        kinds = rng.random(read_len)
        sub = e & (kinds >= indel_err_frac)
        r[sub] = _ACGT[(np.searchsorted(_ACGT, r[sub]) + 1 + rng.integers(0, 3, size=int(sub.sum()))) & 3]
        if indel_err_frac > 0:
            rep = np.ones(read_len, dtype=np.int64)
            rep[e & (kinds < indel_err_frac / 2)] = 0
            rep[e & (kinds >= indel_err_frac / 2) & (kinds < indel_err_frac)] = 2
            r = np.repeat(r, rep)
        names.append(f"L{i}:Pos={p0}:{names_c[int(which[i])]}")
        reads.append(r)
    return names, reads


def write_fastq(path, names, reads, mate: int | None = None, qual_char: bytes = b"5"):
    """reads: 2-D array or list of 1-D arrays."""
    with open(path, "wb") as fh:
        for i, nm in enumerate(names):
            r = reads[i]
            tag = b"" if mate is None else b"\t/%d" % mate
            fh.write(b"@" + nm.encode() + tag + b"\n" + r.tobytes() + b"\n+\n" + qual_char * len(r) + b"\n")


def write_fasta_reads(path, names, reads, mate: int | None = None):
    with open(path, "wb") as fh:
        for i, nm in enumerate(names):
            tag = b"" if mate is None else b"\t/%d" % mate
            fh.write(b">" + nm.encode() + tag + b"\n" + reads[i].tobytes() + b"\n")


_NT4 = np.full(256, 4, dtype=np.uint8)
for _i, _ch in enumerate(b"ACGT"):
    _NT4[_ch] = _i
    _NT4[_ch + 32] = _i


def encode(a: np.ndarray) -> np.ndarray:
    """ASCII -> Kart's 0..3 / 4 codes (nst_nt4_table, reference src/BWT_Index/bntseq.c:40-57)."""
    return _NT4[a]

"""Synthetic reads for bench.py and the hg38-size tests, after the model of the reference's bundled simulator
(/root/reference/wgsim/wgsim.c), generated on the device with torch.  Measurement input only: nothing here is product code.

What wgsim does, and what this module restates (the model and its rates, not its drand48 stream):

  * wgsim_mut_diref (wgsim.c:104-166): the reference becomes TWO haplotypes.  A position mutates with MUT_RATE 0.001; a mutation is an
    indel with INDEL_FRAC 0.15 (half deletions, half insertions), else a substitution by one of the three other bases; a deletion is
    extended base by base with INDEL_EXTEND 0.3 and positions inside a running deletion draw no mutation of their own; an insertion
    holds 1 .. 4 random bases (extended with 0.3, at most 4: they are packed into 8 bits); a mutation is homozygous with probability
    1/3, else on one haplotype picked by a coin.                                                             -> class Haplotypes
  * wgsim_core (wgsim.c:243-391): per pair, outer distance d = round(N(500, 50)), at least the read length; start uniform in the
    contig; mate 0 walks the CHOSEN HAPLOTYPE (a coin per pair) forwards from `pos`, mate 1 backwards from pos + d - 1 and is
    complemented; both reads always hold exactly read-length bases (a read that crosses a deletion reaches further, one that crosses an
    insertion less far); with probability 1/2 the mates swap files; sequencing errors are substitutions only, rate -e, RECURRENT:
    c -> (c + 1) & 3 (wgsim.c:368), so an indel in a read is always a haplotype indel; the quality character is the same for every
    base, (int)(-10 log10(e) + 0.499) + 33; names are "@<pair>_Pos=<start>\\t/<mate>".                         -> gen_reads_device, write_*

A read of this module is a contiguous piece of a MATERIALISED haplotype (the haplotype's own sequence, deletions removed and insertions
in place), which is what wgsim's walk over the annotated reference yields -- except that its backward walk emits an insertion's bases
after the reference base it hangs on (wgsim.c:331-334 under `--i`), which read forwards puts random inserted bases before that base
instead of behind it: the same distribution.  N: wgsim emits N only where the reference has one and drops a pair with more than 5 % of
them in a mate (MAX_N_RATIO); the synthetic genomes here have no N, so reads that overhang an N run of a real assembly (hg38: ~800 gaps)
are modelled as a run of 1 .. 7 N at one end of a read with probability N_OVERHANG per read.

Positions in the names are 1-based offsets into the whole synthetic sequence (decoy + contigs concatenated), not per contig.
"""
import math

import numpy as np
import torch

READ_LEN = 150
DECOY_LEN = 2_000
MUT_RATE, INDEL_FRAC, INDEL_EXTEND = 0.001, 0.15, 0.3     # wgsim.c:99-101
N_OVERHANG = 4e-6
HAP_SEED = 97


def wgsim_quality(err):
    """wgsim.c:259"""
    return ord("I") if err == 0.0 else int(-10.0 * math.log(err) / math.log(10.0) + 0.499) + 33


class Haplotypes:
    """The two haplotypes of `codes` (uint8 0..3 on the device), materialised: self.seq = hap 0 followed by hap 1 in one tensor,
    self.base[h] = where hap h starts in it, plus per haplotype the sorted indel events (reference position, length, kind) and the
    running shift behind each -- enough to send a reference position to its haplotype coordinate (`to_hap`) and to count the indels a
    window of the reference holds (`indels_in`)."""

    def __init__(self, codes, dev, seed=HAP_SEED, mut_rate=MUT_RATE, indel_frac=INDEL_FRAC, indel_extend=INDEL_EXTEND):
        g = torch.Generator(device=dev)
        g.manual_seed(seed)
        L = codes.numel()
        k = int(round((L - 128) * mut_rate))
        pos = torch.unique(torch.randint(DECOY_LEN // 2, L - 64, (k,), generator=g, device=dev, dtype=torch.int64))      # (sorted)
        k = pos.numel()
        u = torch.rand((6, k), generator=g, device=dev, dtype=torch.float64)
        is_indel = u[0] < indel_frac
        is_del = is_indel & (u[1] < 0.5)
        is_ins = is_indel & ~is_del
        hom = u[2] < 0.333333
        which = (u[3] < 0.5).long()
        # lengths: 1 + the number of extensions, each granted with probability indel_extend (an insertion: at most three of them)
        ext = torch.floor(torch.log(u[4].clamp_min(1e-300)) / math.log(indel_extend)).long()
        length = torch.where(is_del, 1 + ext, torch.where(is_ins, 1 + ext.clamp_max(3), torch.zeros_like(ext)))
        length = torch.where(is_del, length.clamp_max(60), length)
        # a position inside a running deletion draws nothing (wgsim.c:117-124)
        end = torch.where(is_del, pos + length, pos)
        covered = torch.zeros(k, dtype=torch.bool, device=dev)
        if k > 1:
            covered[1:] = pos[1:] < torch.cummax(end, 0).values[:-1]
        keep = ~covered
        pos, is_del, is_ins, hom, which, length = pos[keep], is_del[keep], is_ins[keep], hom[keep], which[keep], length[keep]
        k = pos.numel()
        sub_base = (codes[pos] + torch.randint(1, 4, (k,), generator=g, device=dev, dtype=torch.uint8)) & 3
        ins_bases = torch.randint(0, 4, (k, 4), generator=g, device=dev, dtype=torch.uint8)
        self.ref_len = L
        self.n_sites = k
        self.n_indel_sites = int((is_del | is_ins).sum())
        self.ev_pos, self.ev_len, self.ev_del, self.ev_shift, self.hap_len = [], [], [], [], []
        parts = []
        for h in (0, 1):
            mine = hom | (which == h)
            sub = mine & ~is_del & ~is_ins
            src = codes.clone()
            src[pos[sub]] = sub_base[sub]
            ev = mine & (is_del | is_ins)
            ep, el, ed, eb = pos[ev], length[ev], is_del[ev], ins_bases[ev]
            delta = torch.where(ed, -el, el)
            cs = torch.cumsum(delta, 0)
            cs_before = cs - delta
            hl = L + (int(cs[-1]) if cs.numel() else 0)
            # from hap coordinate brk[e] on, the shift is cs[e]: behind an insertion at q that is H(q + 1), behind a deletion H(q + len)
            brk = torch.where(ed, ep + cs_before, ep + 1 + cs)
            out = torch.empty(hl, dtype=torch.uint8, device=dev)
            step = 1 << 27
            for a in range(0, hl, step):
                x = torch.arange(a, min(hl, a + step), device=dev, dtype=torch.int64)
                e = torch.searchsorted(brk, x, right=True) - 1
                shift = torch.where(e >= 0, cs[e.clamp_min(0)], torch.zeros_like(x)) if cs.numel() else torch.zeros_like(x)
                out[a:a + x.numel()] = src[(x - shift).clamp_(0, L - 1)]
                del x, e, shift
            # the inserted bases themselves: behind the base at H(q) = q + cs_before
            ins = ~ed
            if bool(ins.any()):
                hq = (ep + cs_before)[ins]
                n = el[ins]
                for t in range(4):
                    m = n > t
                    out[hq[m] + 1 + t] = eb[ins][m, t]
            del src
            parts.append(out)
            self.ev_pos.append(ep); self.ev_len.append(el); self.ev_del.append(ed); self.ev_shift.append(cs); self.hap_len.append(hl)
        self.base = [0, parts[0].numel()]
        self.seq = torch.cat(parts)
        del parts

    def to_hap(self, h, p):
        """haplotype-h coordinate of reference position p (int64 tensor); a deleted position goes to the first kept one behind it"""
        ep, el, ed, cs = self.ev_pos[h], self.ev_len[h], self.ev_del[h], self.ev_shift[h]
        if ep.numel() == 0:
            return p.clone()
        e = torch.searchsorted(ep, p, right=True) - 1            # the last event at or before p
        ec = e.clamp_min(0)
        has = e >= 0
        q, ln, dl = ep[ec], el[ec], ed[ec]
        inside = has & dl & (p < q + ln)
        p2 = torch.where(inside, q + ln, p)
        at_ins = has & ~dl & (q == p)                            # the base an insertion hangs on keeps the shift BEFORE that insertion
        shift = torch.where(has, cs[ec], torch.zeros_like(p))
        shift = torch.where(at_ins, shift - ln, shift)
        return p2 + shift

    def indels_in(self, h, a, b):
        """number of indel events of haplotype h with reference position in [a, b)"""
        ep = self.ev_pos[h]
        return torch.searchsorted(ep, b, right=False) - torch.searchsorted(ep, a, right=False)


_cache = {}


def haplotypes_of(codes, dev):
    """one Haplotypes per genome tensor, kept until release_haplotypes() (building them for 3.1 Gbp takes a few seconds and 6 GB)"""
    key = (codes.data_ptr(), codes.numel())
    if key not in _cache:
        _cache.clear()
        _cache[key] = Haplotypes(codes, dev)
    return _cache[key]


def release_haplotypes():
    _cache.clear()
    if torch.cuda.is_available():
        torch.cuda.empty_cache()


def _draw_pairs(hp, g, m, dev, err, read_len=READ_LEN):
    """m pairs: (both [m, 2, read_len] codes 0..4 as sequenced from the fragment's two ends -- [:, 0] the forward-strand left end,
    [:, 1] the forward-strand right end, neither complemented yet --, pos, d, n_indel [m, 2])"""
    L = hp.ref_len
    ar = torch.arange(read_len, device=dev)
    d = torch.clamp((torch.randn(m, generator=g, device=dev) * 50 + 500 + 0.5).floor().long(), min=read_len)
    pos = DECOY_LEN + (torch.rand(m, generator=g, device=dev, dtype=torch.float64) * (L - DECOY_LEN - d - 64)).long()
    hap = (torch.rand(m, generator=g, device=dev) < 0.5).long()
    s1 = torch.where(hap == 0, hp.to_hap(0, pos), hp.to_hap(1, pos))
    e2 = torch.where(hap == 0, hp.to_hap(0, pos + d - 1), hp.to_hap(1, pos + d - 1))
    base = torch.where(hap == 0, torch.full_like(hap, hp.base[0]), torch.full_like(hap, hp.base[1]))
    hlen = torch.where(hap == 0, torch.full_like(hap, hp.hap_len[0]), torch.full_like(hap, hp.hap_len[1]))
    s1 = torch.minimum(s1, hlen - read_len)
    e2 = torch.minimum(torch.maximum(e2, torch.full_like(e2, read_len - 1)), hlen - 1)
    left = hp.seq[(base + s1)[:, None] + ar]
    right = hp.seq[(base + e2 - (read_len - 1))[:, None] + ar]
    both = torch.stack([left, right], 1)
    # indels the two windows hold (reference coordinates; a window that crosses a deletion reaches a little further: close enough for a share)
    ni = torch.stack([torch.where(hap == 0, hp.indels_in(0, pos, pos + read_len), hp.indels_in(1, pos, pos + read_len)),
                      torch.where(hap == 0, hp.indels_in(0, pos + d - read_len, pos + d), hp.indels_in(1, pos + d - read_len, pos + d))], 1)
    return both, pos, d, ni


def _sequencing_errors(x, g, err, dev):
    e = torch.rand(x.shape, generator=g, device=dev) < err
    return torch.where(e, (x + 1) & 3, x)                       # recurrent, wgsim.c:368


def _n_overhang(x, g, dev, rate=N_OVERHANG):
    """x [m, len]: with probability `rate` a read gets 1 .. 7 N (code 4) at its start or its end"""
    m, ln = x.shape
    hit = torch.rand(m, generator=g, device=dev, dtype=torch.float64) < rate
    if not bool(hit.any()):
        return x
    rows = torch.nonzero(hit).flatten()
    n = torch.randint(1, 8, (rows.numel(),), generator=g, device=dev)
    at_end = torch.rand(rows.numel(), generator=g, device=dev) < 0.5
    ar = torch.arange(ln, device=dev)
    mask = torch.where(at_end[:, None], ar[None, :] >= (ln - n)[:, None], ar[None, :] < n[:, None])
    sub = x[rows]
    sub[mask] = 4
    x[rows] = sub
    return x


def gen_reads_device(genome_codes, n_pairs, seed, err, dev, want_meta=False):
    """(enc uint8 [2 * n_pairs * 150] codes 0..4, offsets int64) on the device, the reads as the mapper holds them: mate 1 as sequenced,
    mate 2 reverse-complemented (reference src/GetData.cpp:125-135).  want_meta: also (start [n_pairs, 2] 1-based walk starts as wgsim
    names them, stats dict)."""
    hp = haplotypes_of(genome_codes, dev)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    enc = torch.empty(2 * n_pairs * READ_LEN, dtype=torch.uint8, device=dev)
    view = enc.view(n_pairs, 2, READ_LEN)
    starts = torch.empty((n_pairs, 2), dtype=torch.int64, device=dev) if want_meta else None
    stats = {"pairs": n_pairs, "pairs_with_indel": 0, "reads_with_indel": 0, "reads_with_N": 0}
    chunk = 1 << 20
    rc = lambda x: torch.where(x < 4, 3 - x, x).flip(1)
    for s in range(0, n_pairs, chunk):
        m = min(chunk, n_pairs - s)
        both, pos, d, ni = _draw_pairs(hp, g, m, dev, err)
        left, right_fwd = both[:, 0], both[:, 1]
        flip = torch.rand(m, generator=g, device=dev) < 0.5
        # file 1 holds the forward-walking mate, or (flip) the backward-walking one; the mapper reverse-complements what file 2 holds:
        # read 1 forward / read 2 forward, or both on the reverse strand
        r1 = torch.where(flip[:, None], rc(right_fwd), left)
        r2 = torch.where(flip[:, None], rc(left), right_fwd)
        pair = torch.stack([r1, r2], 1)
        pair = _sequencing_errors(pair, g, err, dev)
        pair = _n_overhang(pair.view(2 * m, READ_LEN), g, dev).view(m, 2, READ_LEN)
        view[s:s + m] = pair
        if want_meta:
            a, b = pos + 1, pos + d
            starts[s:s + m, 0] = torch.where(flip, b, a)
            starts[s:s + m, 1] = torch.where(flip, a, b)
        stats["pairs_with_indel"] += int((ni.sum(1) > 0).sum())
        stats["reads_with_indel"] += int((ni > 0).sum())
        stats["reads_with_N"] += int((pair == 4).any(2).sum())
    offsets = torch.arange(0, 2 * n_pairs + 1, device=dev, dtype=torch.int64) * READ_LEN
    if want_meta:
        return enc, offsets, starts, stats
    return enc, offsets


MAX_REC_BYTES = 1 + 9 + 5 + 10 + 3 + 1 + READ_LEN + 3 + READ_LEN + 1     # the longest record write_fastq_pairs() can write


def _decimal(x, width, dev):
    """[m] int64 -> [m, width] ASCII digits with the leading zeros as 0 bytes (dropped when the records are compacted)"""
    pow10 = torch.tensor([10 ** (width - 1 - k) for k in range(width)], device=dev, dtype=torch.int64)
    dig = (x[:, None] // pow10) % 10
    lead = (torch.cumsum(dig != 0, 1) == 0)
    lead[:, -1] = False
    return torch.where(lead, torch.zeros_like(dig), dig + 48).to(torch.uint8)


def _records(idx, start, mate_char, bases, qual, dev, tail):
    """[m, W] byte matrix of FASTQ records "@<idx>_Pos=<start><tail>\\n<bases>\\n+\\n<qual x len>\\n" with 0 bytes where a shorter number
    leaves room; compact with rec[rec != 0]"""
    m, ln = bases.shape
    tail_b = list(tail)
    w_name = 1 + 9 + 5 + 10 + len(tail_b) + 1
    rec = torch.zeros((m, w_name + ln + 3 + ln + 1), dtype=torch.uint8, device=dev)
    rec[:, 0] = 64
    rec[:, 1:10] = _decimal(idx, 9, dev)
    rec[:, 10:15] = torch.tensor(list(b"_Pos="), dtype=torch.uint8, device=dev)
    rec[:, 15:25] = _decimal(start, 10, dev)
    at = 25
    for c in tail_b:
        rec[:, at] = c
        at += 1
    if mate_char is not None:
        rec[:, at - 1] = mate_char
    rec[:, at] = 10
    at += 1
    rec[:, at:at + ln] = bases
    at += ln
    rec[:, at] = 10; rec[:, at + 1] = 43; rec[:, at + 2] = 10
    at += 3
    rec[:, at:at + ln] = qual
    rec[:, at + ln] = 10
    return rec


def write_fastq_pairs(codes, n_pairs, seed, f1, f2, dev, err=0.01):
    """Two FASTQ files of n_pairs records each from gen_reads_device, named and laid out as wgsim writes them
    ("@<pair>_Pos=<start>\\t/<mate>", bases, "+", one quality character throughout): records of varying width.  Assembled on the device
    a million at a time.  Returns the generator's stats (share of pairs / reads that hold a haplotype indel, reads with N)."""
    enc, _, starts, stats = gen_reads_device(codes, n_pairs, seed=seed, err=err, dev=dev, want_meta=True)
    view = enc.view(n_pairs, 2, READ_LEN)
    acgt = torch.tensor(list(b"ACGTN"), dtype=torch.uint8, device=dev)
    comp = torch.tensor(list(b"TGCAN"), dtype=torch.uint8, device=dev)
    q = wgsim_quality(err)
    slab = 1 << 20
    total = 0
    for path, mate in ((f1, 0), (f2, 1)):
        with open(path, "wb") as fh:
            for s in range(0, n_pairs, slab):
                m = min(slab, n_pairs - s)
                idx = torch.arange(s, s + m, device=dev, dtype=torch.int64)
                # the generator holds mate 2 as the mapper does (reverse-complemented): the file holds it as sequenced
                bases = acgt[view[s:s + m, 0, :].long()] if mate == 0 else comp[view[s:s + m, 1, :].flip(1).long()]
                rec = _records(idx, starts[s:s + m, mate], 49 + mate, bases, q, dev, b"\t/1")
                flat = rec[rec != 0]
                total += flat.numel()
                fh.write(memoryview(flat.cpu().numpy()).cast("B"))
    del enc
    stats["fastq_bytes"] = total
    stats["err"] = err
    return stats


def write_long_reads(codes, n_long, read_len, seed, path, dev, err=0.15):
    """One FASTQ file of n_long single-end reads of read_len bases (configs[3]: 7000 at 15 %): a piece of a haplotype (a coin per read)
    from either strand -- what one of wgsim's two mates is --, recurrent substitution errors at rate err, wgsim's names and quality
    character; 20 000 records at a time.  Returns stats (share of reads that hold a haplotype indel)."""
    hp = haplotypes_of(codes, dev)
    g = torch.Generator(device=dev); g.manual_seed(seed)
    acgt = torch.tensor(list(b"ACGTN"), dtype=torch.uint8, device=dev)
    ar = torch.arange(read_len, device=dev)
    L = hp.ref_len
    q = wgsim_quality(err)
    stats = {"reads": n_long, "reads_with_indel": 0, "indels": 0, "fastq_bytes": 0, "err": err}
    with open(path, "wb") as fh:
        for s in range(0, n_long, 20000):
            m = min(20000, n_long - s)
            pos = DECOY_LEN + (torch.rand(m, generator=g, device=dev, dtype=torch.float64) * (L - DECOY_LEN - read_len - 128)).long()
            hap = (torch.rand(m, generator=g, device=dev) < 0.5).long()
            h0 = torch.where(hap == 0, hp.to_hap(0, pos), hp.to_hap(1, pos))
            hlen = torch.where(hap == 0, torch.full_like(hap, hp.hap_len[0]), torch.full_like(hap, hp.hap_len[1]))
            h0 = torch.minimum(h0, hlen - read_len)
            base = torch.where(hap == 0, torch.full_like(hap, hp.base[0]), torch.full_like(hap, hp.base[1]))
            r = hp.seq[(base + h0)[:, None] + ar]
            ni = torch.where(hap == 0, hp.indels_in(0, pos, pos + read_len), hp.indels_in(1, pos, pos + read_len))
            stats["reads_with_indel"] += int((ni > 0).sum()); stats["indels"] += int(ni.sum())
            flip = torch.rand(m, generator=g, device=dev) < 0.5
            r = torch.where(flip[:, None], (3 - r).flip(1), r)
            r = _sequencing_errors(r, g, err, dev)
            r = _n_overhang(r, g, dev)
            idx = torch.arange(s, s + m, device=dev, dtype=torch.int64)
            start = torch.where(flip, pos + read_len, pos + 1)
            rec = _records(idx, start, None, acgt[r.long()], q, dev, b"\t/1")
            flat = rec[rec != 0]
            stats["fastq_bytes"] += flat.numel()
            fh.write(memoryview(flat.cpu().numpy()).cast("B"))
    return stats


# ---- prefixes and slices of such files (records of varying width: a prefix is a number of lines, not a byte range) ---------------
def copy_records(src, dst, n_records, skip_records=0, lines_per_record=4):
    """the records [skip, skip + n) of FASTQ file src into dst; returns the number of records written"""
    want_skip, want = skip_records * lines_per_record, n_records * lines_per_record
    seen = 0
    with open(src, "rb") as fi, open(dst, "wb") as fo:
        while want > 0:
            buf = fi.read(32 << 20)
            if not buf:
                break
            a = np.frombuffer(buf, dtype=np.uint8)
            nl = np.flatnonzero(a == 10)
            lo = 0
            if want_skip > 0:
                if len(nl) <= want_skip - 1:
                    want_skip -= len(nl)
                    continue
                lo = int(nl[want_skip - 1]) + 1
                nl = nl[want_skip:]
                want_skip = 0
            if len(nl) >= want:
                hi = int(nl[want - 1]) + 1
                fo.write(buf[lo:hi])
                seen += want
                want = 0
            else:
                fo.write(buf[lo:])
                seen += len(nl)
                want -= len(nl)
    return seen // lines_per_record


def split_records(src, dsts, per, lines_per_record=4):
    """consecutive slices of `per` records each"""
    for k, dst in enumerate(dsts):
        copy_records(src, dst, per, skip_records=k * per, lines_per_record=lines_per_record)


def record_starts(path):
    """byte offsets of every FASTQ record's sequence line and its length (small files only: the file is read whole)"""
    data = np.fromfile(path, dtype=np.uint8)
    nl = np.flatnonzero(data == 10)
    seq_start = nl[0::4] + 1
    seq_len = nl[1::4] - seq_start
    return seq_start, seq_len
